"""Semantic point-cloud export -- mirror of ``generate_point_cloud``
(``crop_nerf/fruit_nerf/export/exporter_utils_nerfacto.py:83-227``): random train rays -> model forward ->
``point = o + d * depth`` kept where ``semantics_colormap[:, 0] > 0`` (``:156-166``) and inside the optional oriented
box (``:168-174``, folded into the same mask) -> accumulate
until ``num_points``.  Pixel draws (``cn_pixel_sample``), mask, point computation and compaction
(``cn_pointcloud_compact_calls``) run on the device, several of the reference's calls per launch; kept points leave the
device once.  The statistical outlier removal (``:194-199``, open3d ``remove_statistical_outlier(nb_neighbors=20,
std_ratio)``, on by default) runs on the device too (``ops.statistical_outlier_mask``: uniform-grid k-nearest search,
``cn_knn_mean_distance``), and so do the normals (``:200-225``): ``cn_estimate_normals`` restates open3d's
``estimate_normals()`` defaults (30 nearest neighbours, covariance, smallest eigenvector), the re-orientation against the
kept rays' view directions follows.  open3d is not imported anywhere on this path."""

from __future__ import annotations

import sys
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from ... import ops


DEFAULT_LAUNCH_RAYS = 1 << 20  # calls per launch sequence = this // the call size (512 of the reference's 2 048-ray calls)


def generate_point_cloud(pipeline, num_points: int = 1000000, remove_outliers: bool = True,
                         estimate_normals: bool = False, reorient_normals: bool = False, rgb_output_name: str = "rgb",
                         depth_output_name: str = "depth", normal_output_name: Optional[str] = None, crop_obb=None,
                         std_ratio: float = 10.0, only_semantics: bool = True, max_batches: Optional[int] = None,
                         use_graph: bool = True, launch_rays: Optional[int] = DEFAULT_LAUNCH_RAYS,
                         stats: Optional[dict] = None, sort_rays: bool = True) -> Dict[str, np.ndarray]:
    """``launch_rays`` (extension): the reference's call size (``train_num_rays_per_batch``: 2 048 in its own exporter, 32 768
    upstream) bounds ITS memory and fixes its stopping rule -- the cloud is every kept point of calls 0 .. c*, c* = the first
    call at which the count reaches ``num_points``.  Here K = ``launch_rays // rays_per_call`` calls' pixel draws go through ONE
    sampler + render + compaction launch sequence; the draws come from a counter-based stream (``cn_pixel_sample``) and the
    append stops at the same call boundary (``cn_pointcloud_compact_calls``), so the cloud does not depend on K (tested).
    ``launch_rays=None`` (or below the call size): one call per launch, as the reference loops.  ``stats``: filled with the
    calls / rays rendered.

    ``sort_rays`` (extension, with several calls per launch): the launch's rays are rendered in (camera, pixel Morton) order and
    their outputs put back in draw order before the compaction.  The draws are random pixels of random cameras, whose field
    gathers miss the L2 (1.77 ms per 65 536 rays against 0.73 ms for an image's); inside a 2^20-ray launch the sorted rays are
    neighbours again (0.93 ms).  A ray's result does not depend on its neighbours: the cloud is the same, point for point."""
    model, dm = pipeline.model, pipeline.datamanager
    # several ranks (one per GPU): every rank collects its share of the target from its own random rays; the shares are
    # concatenated with one variable-length all-gather (counts, then padded rows) before the outlier pass
    from ...distributed import all_gather_points, world as _world

    rank, world_size = _world()
    if world_size > 1:
        num_points = -(-num_points // world_size)
    rays_per_call = int(dm.config.train_num_rays_per_batch)
    K = max(1, int(launch_rays) // rays_per_call) if launch_rays else 1
    if max_batches is not None:
        K = max(1, min(K, int(max_batches)))
    # The reference reads the kept-point count back after every call (``while num_points < total``).  Here up to ``lookahead``
    # launches are enqueued before one read-back of the running count; a launch issued after the target was reached appends
    # nothing (its ray limit is 0), so the result is exactly the reference's, without a host round trip per call.
    lookahead = max(1, 16 // K) if K > 1 else 16
    cap = int(num_points + rays_per_call + 64)
    cams = dm.cameras
    n_cam, H, W = len(cams), int(cams.height), int(cams.width)
    dev = model.device
    call_no = torch.zeros(1, dtype=torch.int64, device=dev)  # first call of the next launch (device: a graph advances it)
    state = {"buffers": None}

    sort_launch = bool(sort_rays) and K > 1

    def one_launch():
        idx = ops.pixel_sample(dm.export_seed, call_no, K, rays_per_call, n_cam, H, W)
        perm = ops.ray_sort_permutation(idx, H, W) if sort_launch else None
        ray_bundle = cams.generate_rays(idx if perm is None else idx[perm])  # data/fruit_datamanager.py:188-197
        outputs = model(ray_bundle)
        for name in (rgb_output_name, depth_output_name):
            if name not in outputs:  # :133-142
                print(f"Could not find {name} in the model outputs; choose one of: {list(outputs.keys())}", file=sys.stderr)
                sys.exit(1)
        cmap = outputs["semantics_colormap"] if only_semantics else torch.ones_like(outputs["rgb"])
        if crop_obb is not None:  # :168-174: cropped points do not count towards num_points
            inside = crop_obb.within(ray_bundle.origins + ray_bundle.directions * outputs[depth_output_name])
            cmap = cmap * inside[:, None].to(cmap.dtype)
        per_ray = [ray_bundle.origins, ray_bundle.directions, outputs[depth_output_name], outputs[rgb_output_name], cmap]
        if perm is not None:  # back into draw order: the compaction cuts the cloud at a CALL boundary
            per_ray = [torch.empty_like(t).index_copy_(0, perm, t) for t in per_ray]
        o_, d_, depth_, rgb_, cmap_ = per_ray
        state["buffers"] = ops.pointcloud_compact_calls(o_, d_, depth_, rgb_, cmap_.contiguous(), rays_per_call, num_points,
                                                        cap, state["buffers"])
        call_no.add_(K)

    kept = 0
    launches = 0
    with torch.no_grad():
        # A 2 048-ray call is ~15 small launches: that loop is bound by the host.  After three eager launches (they count, and
        # they run every first-call initialisation) the launch sequence is captured into a HIP graph -- the call counter is
        # device memory, every intermediate lives in the graph's memory pool -- and replayed.
        graph = None
        replay = None
        # (a launch of 2^18 rays or more is tens of milliseconds of device work: nothing to gain from a replay, and the sort's
        #  temporaries stay out of a graph pool)
        if use_graph and max_batches is None and K * rays_per_call < (1 << 18):
            for j in range(3):
                one_launch()
            launches = 3
            kept = int(state["buffers"][3].item())
            if kept < num_points:
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):  # records, runs nothing: counters and buffers keep their values
                        one_launch()
                    replay = graph.replay
                except Exception as e:  # capture is an optimisation: fall back to eager launches, loudly
                    print(f"[generate_point_cloud] HIP graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                    graph, replay = None, None
                    torch.cuda.synchronize()
        done = kept >= num_points
        while not done:
            group = lookahead
            if max_batches is not None:
                group = min(lookahead, -(-(max_batches - launches * K) // K))
            for j in range(group):
                if replay is not None:
                    replay()
                else:
                    one_launch()
            launches += group
            kept = int(state["buffers"][3].item())  # the one synchronisation per group
            done = kept >= num_points or (max_batches is not None and launches * K >= max_batches)
    dm.train_count += launches * K
    buffers = state["buffers"]
    pts, cols, dirs = buffers[0], buffers[1], buffers[2]
    n = min(kept, cap)
    pts, cols, dirs = pts[:n], cols[:n], dirs[:n]
    if stats is not None:
        stats.update(calls=launches * K, rays=launches * K * rays_per_call, launches=launches, calls_per_launch=K,
                     rays_per_call=rays_per_call, graph=replay is not None, kept=n)
    if world_size > 1:
        rows = all_gather_points(torch.cat([pts, cols, dirs], dim=-1).contiguous())
        pts, cols, dirs = rows[:, 0:3].contiguous(), rows[:, 3:6].contiguous(), rows[:, 6:9].contiguous()
    if remove_outliers and pts.shape[0] > 0:
        # on the device, before the points leave it.  open3d works on the float64 cloud; the search here is float32
        # (positions of a +-1 scene: mean neighbour distances agree to ~1e-5 relative, so only points sitting on the
        # threshold can flip)
        keep = ops.statistical_outlier_mask(pts.contiguous(), 20, std_ratio)
        pts, cols, dirs = pts[keep], cols[keep], dirs[keep]
    result: Dict[str, np.ndarray] = {}
    if estimate_normals:  # :203-212
        if normal_output_name is not None:
            print("Cannot estimate normals and use normal_output_name at the same time", file=sys.stderr)
            sys.exit(1)
        # open3d's estimate_normals() at its defaults, on the device (cn_estimate_normals: the 30 nearest points' covariance,
        # eigenvector of its smallest eigenvalue), before the cloud leaves it
        normals, degenerate = ops.estimate_normals(pts.contiguous(), 30)
        if reorient_normals:  # :219-225 (the float32 round trip is the reference's)
            normals, _ = ops.reorient_normals(normals, dirs)
        result["normals"] = normals.cpu().numpy()
        result["degenerate_normals"] = int(degenerate.sum())
    elif normal_output_name is not None:
        # the reference reads a "normals" model output here (:213-217); FruitModel predicts none (predict_normals is never set
        # in its configs), so its own `outputs[normal_output_name]` raises KeyError as well
        raise KeyError(f"the model has no output {normal_output_name!r}: use --normal-method open3d")
    result.update(points=pts.double().cpu().numpy(), colors=cols.double().cpu().numpy(), view_directions=dirs.cpu().numpy())
    return result


def render_trajectory(pipeline, cameras, rgb_output_name: str, depth_output_name: str,
                      rendered_resolution_scaling_factor: float = 1.0, disable_distortion: bool = False,
                      return_rgba_images: bool = False) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """``exporter_utils_nerfacto.py:230-287``: every camera's full image through ``get_outputs_for_camera_ray_bundle``;
    rgb and depth images as numpy arrays.  ``disable_distortion`` is accepted and has nothing to disable (the cameras of
    this path are undistorted pinholes).  ``return_rgba_images``: rgb with the accumulation as alpha (nerfstudio's
    ``get_rgba_image``)."""
    images, depths = [], []
    cameras.rescale_output_resolution(rendered_resolution_scaling_factor)
    cameras = cameras.to(pipeline.model.device)
    for camera_idx in range(cameras.size):
        camera_ray_bundle = cameras.generate_rays(camera_idx, keep_shape=True)
        with torch.no_grad():
            outputs = pipeline.model.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
        for flag, name in (("--rgb_output_name", rgb_output_name), ("--depth_output_name", depth_output_name)):
            if name not in outputs:  # :269-278
                print(f"Could not find {name} in the model outputs\nPlease set {flag} to one of: {list(outputs.keys())}",
                      file=sys.stderr)
                sys.exit(1)
        image = outputs[rgb_output_name]
        if return_rgba_images:
            image = torch.cat([image, outputs["accumulation"]], dim=-1)
        images.append(image.cpu().numpy())
        depths.append(outputs[depth_output_name].cpu().numpy())
    return images, depths


def collect_camera_poses_for_dataset(dataset, camera_optimizer=None) -> List[Dict[str, Any]]:
    """``exporter_utils_nerfacto.py:290-334``: one ``{"file_path", "transform"}`` per camera of ``dataset``; ``transform``
    is the stored 3 x 4 camera-to-world, or -- with a camera optimiser -- ``camera_optimizer.apply_to_camera`` of it."""
    if dataset is None:
        return []
    cameras = dataset.cameras
    image_filenames = dataset.image_filenames
    frames: List[Dict[str, Any]] = []
    for idx in range(len(cameras)):
        if camera_optimizer is None:
            transform = cameras.camera_to_worlds[idx].tolist()
        else:
            camera = cameras[idx:idx + 1]
            assert camera.metadata is not None
            camera.metadata["cam_idx"] = idx
            transform = camera_optimizer.apply_to_camera(camera).tolist()[0]
        frames.append({"file_path": str(image_filenames[idx]), "transform": transform})
    return frames


def collect_camera_poses(pipeline) -> Tuple[List[Dict[str, Any]], List[Dict[str, Any]]]:
    """``exporter_utils_nerfacto.py:337-357``: training frames with the optimised poses, eval frames with the original
    ones."""
    train_dataset = pipeline.datamanager.train_dataset
    assert train_dataset is not None
    eval_dataset = getattr(pipeline.datamanager, "eval_dataset", None)
    camera_optimizer = getattr(pipeline.model, "camera_optimizer", None)
    train_frames = collect_camera_poses_for_dataset(train_dataset, camera_optimizer)
    eval_frames = collect_camera_poses_for_dataset(eval_dataset)
    return train_frames, eval_frames

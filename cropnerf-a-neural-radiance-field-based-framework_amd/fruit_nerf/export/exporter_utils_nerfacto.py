"""Semantic point-cloud export -- mirror of ``generate_point_cloud``
(``crop_nerf/fruit_nerf/export/exporter_utils_nerfacto.py:83-227``): random train rays -> model forward ->
``point = o + d * depth`` kept where ``semantics_colormap[:, 0] > 0`` (``:156-166``) and inside the optional oriented
box (``:168-174``, folded into the same mask) -> accumulate
until ``num_points``.  Mask, point computation and compaction run in ``cn_pointcloud_compact``; kept points leave the
device once.  The statistical outlier removal (``:194-199``, open3d ``remove_statistical_outlier(nb_neighbors=20,
std_ratio)``, on by default) runs on the device too (``ops.statistical_outlier_mask``: uniform-grid k-nearest search,
``cn_knn_mean_distance``).  Normal estimation / re-orientation (``:200-225``) stays open3d CPU post-processing and is
applied only when open3d is importable."""

from __future__ import annotations

import sys
from typing import Dict, Optional

import numpy as np
import torch

from ... import ops


def generate_point_cloud(pipeline, num_points: int = 1000000, remove_outliers: bool = True,
                         estimate_normals: bool = False, reorient_normals: bool = False, rgb_output_name: str = "rgb",
                         depth_output_name: str = "depth", normal_output_name: Optional[str] = None, crop_obb=None,
                         std_ratio: float = 10.0, only_semantics: bool = True, max_batches: Optional[int] = None,
                         use_graph: bool = True) -> Dict[str, np.ndarray]:
    model, dm = pipeline.model, pipeline.datamanager
    # several ranks (one per GPU): every rank collects its share of the target from its own random rays; the shares are
    # concatenated with one variable-length all-gather (counts, then padded rows) before the outlier pass
    from ...distributed import all_gather_points, world as _world

    rank, world_size = _world()
    if world_size > 1:
        num_points = -(-num_points // world_size)
    # The reference reads the kept-point count back after every 2048-ray call (``while num_points < total``).  Here up to
    # ``lookahead`` calls are enqueued before one read-back of their running counts; the cloud is then cut at the count of
    # the first call that reached ``num_points``, so the result is exactly the reference's (calls append in order), without
    # a host round trip per call.
    lookahead = 16
    rays_per_call = dm.config.train_num_rays_per_batch
    cap = int(num_points + (lookahead + 4) * rays_per_call)
    state = {"buffers": None}

    def one_call(next_rays):
        ray_bundle, _ = next_rays(0)
        outputs = model(ray_bundle)
        for name in (rgb_output_name, depth_output_name):
            if name not in outputs:  # :133-142
                print(f"Could not find {name} in the model outputs; choose one of: {list(outputs.keys())}", file=sys.stderr)
                sys.exit(1)
        cmap = outputs["semantics_colormap"] if only_semantics else torch.ones_like(outputs["rgb"])
        if crop_obb is not None:  # :168-174: cropped points do not count towards num_points
            inside = crop_obb.within(ray_bundle.origins + ray_bundle.directions * outputs[depth_output_name])
            cmap = cmap * inside[:, None].to(cmap.dtype)
        state["buffers"] = ops.pointcloud_compact(ray_bundle.origins, ray_bundle.directions, outputs[depth_output_name],
                                                  outputs[rgb_output_name], cmap.contiguous(), cap, state["buffers"])

    kept = 0
    batches = 0
    with torch.no_grad():
        history = torch.zeros(lookahead, dtype=torch.int64, device=model.device)
        # One call is ~15 small launches for 2048 rays: the loop is bound by the host.  After three eager calls (they
        # count, and they run every first-call initialisation) the call is captured into a HIP graph -- pixel indices from
        # the device generator, every intermediate in the graph's memory pool -- and replayed.
        graph = None
        replay = None
        if use_graph and max_batches is None:
            for j in range(3):
                one_call(dm.next_train_device)
                history[j].copy_(state["buffers"][3].reshape(()))
            batches = 3
            try:
                graph = torch.cuda.CUDAGraph()
                graph.register_generator_state(dm.device_generator)  # the pixel sampler's own (per-rank) random stream
                with torch.cuda.graph(graph):
                    one_call(dm.next_train_device)
                replay = graph.replay
                dm.train_count -= 1  # the capture pass enqueued nothing
            except Exception as e:  # capture is an optimisation: fall back to eager launches, loudly
                print(f"[generate_point_cloud] HIP graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph, replay = None, None
                torch.cuda.synchronize()
            counts = history[:3].tolist()
            kept = next((c for c in counts if c >= num_points), counts[-1])
        done = kept >= num_points
        while not done:
            group = lookahead if max_batches is None else min(lookahead, max_batches - batches)
            for j in range(group):
                if replay is not None:
                    replay()
                    dm.train_count += 1
                else:
                    one_call(dm.next_train)
                history[j].copy_(state["buffers"][3].reshape(()))
            counts = history[:group].tolist()  # the one synchronisation per group
            batches += group
            kept = counts[-1]
            for c in counts:
                if c >= num_points:
                    kept, done = c, True
                    break
            if max_batches is not None and batches >= max_batches:
                done = True
    buffers = state["buffers"]
    pts, cols, dirs, count = buffers
    n = min(kept, cap)
    pts, cols, dirs = pts[:n], cols[:n], dirs[:n]
    if world_size > 1:
        rows = all_gather_points(torch.cat([pts, cols, dirs], dim=-1).contiguous())
        pts, cols, dirs = rows[:, 0:3].contiguous(), rows[:, 3:6].contiguous(), rows[:, 6:9].contiguous()
    if remove_outliers and pts.shape[0] > 0:
        # on the device, before the points leave it.  open3d works on the float64 cloud; the search here is float32
        # (positions of a +-1 scene: mean neighbour distances agree to ~1e-5 relative, so only points sitting on the
        # threshold can flip)
        keep = ops.statistical_outlier_mask(pts.contiguous(), 20, std_ratio)
        pts, cols, dirs = pts[keep], cols[keep], dirs[keep]
    points = pts.double().cpu().numpy()
    colors = cols.double().cpu().numpy()
    view_dirs = dirs.cpu().numpy()
    result = {"points": points, "colors": colors, "view_directions": view_dirs}
    if estimate_normals:
        try:
            import open3d as o3d  # noqa: F401
        except ImportError:
            result["note"] = "open3d not installed: normal estimation skipped"
            return result
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(points)
        pcd.colors = o3d.utility.Vector3dVector(colors)
        if estimate_normals:
            pcd.estimate_normals()
            result["normals"] = np.asarray(pcd.normals)
        result.update(points=np.asarray(pcd.points), colors=np.asarray(pcd.colors), view_directions=view_dirs)
    return result

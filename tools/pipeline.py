"""BASELINE.json configs[4] on one GPU, on the analytic plant: train -> semantic point-cloud export -> segmenter
(super-clusters + k-means sub-clusters) -> NeRF projection of every (super-cluster, camera, sub-cluster) -> merger (image
stage on the device projections, affinity graph, partition, count) and the depth-based projection.  Reports the seconds of each stage and the fruit count (super-clusters) against the number of
bolls of the closed-form plant.

    python tools/pipeline.py [--iters 2000] [--res 200] [--side 800] [--views 8]

BASELINE.json configs[4] proper -- all plants concurrently, one per GPU -- is N independent REPLICAS of this pipeline
(the path does not shard across plants: no collective, no process group):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/pipeline.py [...]

Every rank binds to GPU LOCAL_RANK, takes plant ``--plant + RANK`` (its own training seed / batches) and prints its own
JSON line; nothing is exchanged.  ``CROPNERF_REHEARSE_ON_ONE_GPU=1`` puts every replica on cuda:0 (tests on a one-GPU box).
"""
import argparse, json, os, sys, tempfile, time
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from cropnerf_amd import synthetic
from cropnerf_amd.fruit_nerf.export.exporter_utils import sample_volume
from cropnerf_amd.fruit_nerf.fruit_nerf import Semantics
from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as DP
from cropnerf_amd.segmentation import segmenter as SG
from cropnerf_amd.rays import Cameras
import fit_scene


def clock():
    torch.cuda.synchronize()
    return time.perf_counter()


def main(a):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("CROPNERF_REHEARSE_ON_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)  # replicas: one plant per GPU, no process group is ever created
    plant = a.plant + rank
    out = {"plant": plant, "replica": rank, "replicas": world, "device": f"cuda:{local}"}
    t = clock()
    fit, pipe, (cams_all, images, masks, held) = fit_scene.fit(a.iters, a.res, seed=plant, device=f"cuda:{local}")
    out["train_s"] = round(clock() - t, 2)
    out["held_out_psnr"], out["held_out_fruit_iou"] = fit["held_out_psnr_mean"], fit["held_out_fruit_iou_mean"]
    model, dm = pipe.model, pipe.datamanager
    model.eval()
    model.config.matrix_precision = a.matrix_precision  # eval renders only; training stays exact fp32
    out["matrix_precision"] = a.matrix_precision

    # --- ns-export semantic point cloud (scripts/exporter.py:80-133): side^2 orthographic rays x side samples
    t = clock()
    lo, hi = (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)
    dm.config.eval_num_rays_per_batch = max(512, (1 << 24) // a.side)
    model.setup_inference(render_rgb=True, num_inference_samples=a.side)
    n_rays = dm.setup_inference(num_points=a.side, aabb=(lo, hi))
    model.test_mode, dm.train_count = "export", 0  # the exporter is a fresh process in the reference
    # the export field runs without the scene contraction the model trained with, so its positions are world / 2; the
    # exporter's scale(1 / dataparser scale) . scale(2) (exporter_utils.py:190-191) brings them back
    pcds = sample_volume(pipe, n_rays, sem_thresh=a.sem_thresh, den_thresh=a.den_thresh,
                         transform_json={"scale": 1.0, "transform": np.eye(4)[:3].tolist()})
    model.test_mode = "val"
    out["export_s"] = round(clock() - t, 2)
    out["export_samples"] = n_rays * a.side
    out["export_kept"] = {k: len(v["points"]) for k, v in pcds.items()}
    fruit = pcds["semantic_colormap"]["points"]
    tree = pcds["density"]["points"]

    # --- segmenter.py: super-clusters -> k-means sub-clusters -> all_super_cluster_info
    t = clock()
    vx = 1.0 / a.side  # half a sample spacing of the re-scaled cloud: eps = 20 vx = 10 spacings, min_points = 30 (segmenter.py:76-77)
    SG.get_super_clusters(fruit, vx)
    out["super_clusters_s"] = round(clock() - t, 3)  # voxel down-sample + DBSCAN + outlier removal (device)
    t = clock()
    info = SG.process_and_save_all(fruit, k=2, vx_size=vx)  # the same again + k-means per super-cluster (Lloyd iterations on the device)
    out["segment_s"] = round(clock() - t, 3)
    out["fruit_count"], out["bolls"] = len(info), len(synthetic.BOLLS)
    centres = [np.concatenate(list(sc["pcd"].values())).mean(0) for sc in info]
    out["centres"] = [[round(float(x), 3) for x in c] for c in centres]
    out["cluster_points"] = [int(sum(len(p) for p in sc["pcd"].values())) for sc in info]
    out["centre_error"] = [round(float(min(np.linalg.norm(c - np.asarray(b[0])) for b in synthetic.BOLLS)), 4) for c in centres]

    # --- semantic_projection.py: NeRF projection of every (super-cluster, camera, sub-cluster) job
    cams = dm.cameras
    view_ids = list(range(0, len(cams), max(1, len(cams) // a.views)))[: a.views]

    sel = torch.tensor(view_ids, device=cams.fx.device)

    class _Dataset:
        cameras = Cameras(cams.camera_to_worlds[sel], cams.fx[sel], cams.fy[sel], cams.cx[sel], cams.cy[sel], cams.height, cams.width)
        metadata = {"semantics": Semantics()}

    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context

    t = clock()
    with background_color_override_context(torch.zeros(3)):  # scripts/semantic_projection.py:169
        if a.png_dir:  # the reference's file tree, written by worker threads behind the GPU
            model.get_outputs_for_projections(_Dataset, None, pcd_data=info, output_root=a.png_dir, save=True)
        run = model.get_outputs_for_projections(_Dataset, None, pcd_data=info, save=False, return_run=True)
    out["projection_s"] = round(clock() - t, 3)
    out["projection_jobs"] = int(run.stats["jobs"])
    out["projection_rays"] = int(run.stats["rays"])
    out["projection_batches"] = len(run.batches)

    # --- merger.py: per sub-cluster and camera the un-occluded / visible areas and the instance label under the visible part
    # (image stage, :219-333), affinity + graph partition (:335-355, :26-74) -> fruit count.  The projections go in straight
    # from device memory; the instance-label frames (GroundedSAM's in the reference) are the analytic plant's own.
    from cropnerf_amd.segmentation import merger as MG

    t = clock()
    pc = _Dataset.cameras.to(model.device)
    labels = torch.stack([synthetic.analytic_instance_labels(rb.origins, rb.directions)
                          for rb in (pc.generate_rays(i, keep_shape=True) for i in range(len(pc)))])
    props, visible_jobs = [], 0
    for i_sc in range(len(info)):
        k = np.asarray(info[i_sc]["aabb"]).shape[0]
        wo, vis = run.images_u8(i_sc, k)
        visible_jobs += int((vis.flatten(2) > 0).any(-1).sum())
        props.append(MG.process_super_cluster(wo, vis, labels, binary_thresh=a.binary_threshold, frame_sampling_interval=1,
                                              device=model.device))
    total, node_labels = MG.count_fruit(props, a.graph_partition)
    out["merger_s"] = round(clock() - t, 3)
    out["projection_visible_jobs"] = visible_jobs
    out["merged_fruit_count"] = int(total)
    out["merger_labels"] = [[int(v) for v in l] for l in node_labels]

    # --- depth_based_semantic_projection.py: z-buffer splat of the density cloud (occluder) + the sub-clusters
    t = clock()
    H = W = a.res
    focal = float(cams.fx[0])
    n_depth = 0
    for v in view_ids:
        c2w = np.eye(4)
        c2w[:3] = cams.camera_to_worlds[v].cpu().numpy()
        r = DP.project_and_save_super_clusters(c2w, info, tree, tree, None, intrinsics=(focal, focal, W / 2.0, H / 2.0),
                                               height=H, width=W)
        n_depth += len(r)
    out["depth_projection_s"] = round(clock() - t, 3)
    out["depth_projection_jobs"] = n_depth
    out["total_s"] = round(sum(out[k] for k in ("train_s", "export_s", "segment_s", "projection_s", "merger_s", "depth_projection_s")), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--res", type=int, default=200)
    ap.add_argument("--side", type=int, default=800)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--sem-thresh", type=float, default=3.0)
    ap.add_argument("--den-thresh", type=float, default=70.0)
    ap.add_argument("--matrix-precision", choices=["fp32", "split_bf16", "f16"], default="fp32")
    ap.add_argument("--png-dir", default="", help="also write the projection PNG tree (the reference's artefact) there")
    ap.add_argument("--binary-threshold", type=int, default=100)  # merger.py:375
    ap.add_argument("--graph-partition", default="clique")  # merger.py:373
    ap.add_argument("--plant", type=int, default=0, help="first plant id (rank r of a replicated launch takes plant + r)")
    main(ap.parse_args())

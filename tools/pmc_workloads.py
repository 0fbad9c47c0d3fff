"""One secondary workload of bench.py, run once from start to finish so that rocprofv3 --pmc counters of the whole process
divide cleanly by the units it processed (tools/collect_pmc_workloads.sh runs it under one counter group at a time;
tools/summarise_pmc_workloads.py folds the CSVs into profiles/<tag>_pmc_workloads.json, which bench.py reads for the
`roofline.traffic` of its secondary lines).

    python3 tools/pmc_workloads.py eval_image | projection | export_c4 | dense_export | proposal | render48 | train   [env TRAIN_RAYS, TRAIN_FIELD_SAMPLES]

Prints ONE line `PMC_UNITS {"workload": ..., "units": N, "unit": "..."}`.  Same scene, cameras and call shapes as bench.py."""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cropnerf_amd import config as PC, ops, synthetic  # noqa: E402
from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig  # noqa: E402
from cropnerf_amd.fruit_nerf.fruit_nerf import Semantics, background_color_override_context  # noqa: E402
from cropnerf_amd.fruit_nerf.fruit_pipeline import FruitPipeline, FruitPipelineConfig  # noqa: E402
from cropnerf_amd.rays import Cameras, SceneBox  # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda", 0)
H = W = bench.H
cfg = PC.FruitNerfModelConfig()
c2w, intr = synthetic.orbit_cameras(bench.NUM_CAMERAS, height=H, width=W, focal=bench.FOCAL)
cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], H, W)
box = SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))


def scene_params():
    _, _, _, params, *_ = bench.build_scene(dev)
    p2 = {k: v.clone() for k, v in params.items()}
    p2["field.mlp_base_mlp.layers.1.bias"][0] += 4.0
    p2["field.field_head_semantics.net.bias"] += 3.0
    return p2


def pipeline(rays_per_batch, mode, params):
    return FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(rays_per_batch, rays_per_batch), cfg), dev, cams, box,
                         test_mode=mode, params=params)


units, unit = 0, ""
if what == "eval_image":
    m = pipeline(2048, "test", scene_params()).model
    for i in range(3):
        m.get_outputs_for_camera_ray_bundle(cams.to(dev).generate_rays(i, keep_shape=True))
    units, unit = 3 * H * W, "rays"
elif what == "projection":
    m = pipeline(2048, "test", scene_params()).model
    rng = np.random.default_rng(5)
    boxes = []
    for _ in range(8):
        c, half = rng.uniform(-0.3, 0.3, 3), rng.uniform(0.015, 0.025, 3)
        ax = int(rng.integers(0, 3))
        lo, hi = c - half, c + half
        a, b = hi.copy(), lo.copy()
        a[ax] = b[ax] = c[ax]
        boxes.append({"aabb": np.stack([np.stack([lo, a]), np.stack([b, hi])]).astype(np.float32)})
    sel = torch.arange(0, bench.NUM_CAMERAS, bench.NUM_CAMERAS // 16)[:16]

    class DS:
        cameras = Cameras(c2w[sel], intr[sel, 0], intr[sel, 1], intr[sel, 2], intr[sel, 3], H, W)
        metadata = {"semantics": Semantics()}

    with background_color_override_context(torch.zeros(3)):
        run = m.get_outputs_for_projections(DS, None, pcd_data=boxes, save=False, return_run=True)
    units, unit = 2 * int(run.stats["rays"]), "rays (both passes)"
elif what == "export_c4":
    from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import generate_point_cloud

    p = scene_params()
    p["field.field_head_semantics.net.bias"] += float(os.environ.get("C4_BIAS", "-0.9351"))  # ~3 % of the rays kept (bench.py)
    st = {}
    generate_point_cloud(pipeline(2048, "test", p), num_points=100_000, remove_outliers=False, stats=st)
    units, unit = int(st["rays"]), "rays"
elif what == "dense_export":
    from cropnerf_amd.fruit_nerf.export.exporter_utils import sample_volume

    pipe = pipeline(512, "export", scene_params())
    pipe.datamanager.config.eval_num_rays_per_batch = 512
    pipe.model.setup_inference(True, 3000)
    n_rays = pipe.datamanager.setup_inference(((-1, -1, -1 + .318), (1, 1, 1 + .318)), 128)
    sample_volume(pipe, n_rays, transform_json={"scale": 1.0, "transform": None}, capacity=1 << 24)
    units, unit = n_rays * 3000, "field samples"
elif what == "render48":
    # the default method's field pass on image-coherent C2 batches at 48 samples per ray (proposal bins), composited: the kernel
    # whose schedule CN_SPLIT_PACK switches (two rays per three half-steps / one ray per two)
    cfgb, fspec, pspecs, params, fh, dh, c2w_b, intr_b = bench.build_scene(dev)
    batches = bench.make_batches(ops, c2w_b, intr_b, 0, 1)
    scene_c = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=True)
    for b in batches:
        bins = ops.proposal_sample(dh, scene_c, *b[:4], cfgb.num_proposal_samples_per_ray, 48)["euclidean_bins"]
        ops.render_rays(fh, scene_c, ops.render_opts(48, image_width=W, pixel_start=b[5]), *b[:4], bins=bins)
    units, unit = len(batches) * bench.R, "rays"
elif what == "proposal":
    cfgb, fspec, pspecs, params, fh, dh, c2w_b, intr_b = bench.build_scene(dev)
    batches = bench.make_batches(ops, c2w_b, intr_b, 0, 1)
    scene_c = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=True)
    for b in batches:
        ops.proposal_sample(dh, scene_c, *b[:4], cfgb.num_proposal_samples_per_ray, bench.S)
    units, unit = len(batches) * bench.R, "rays"
elif what == "train":
    from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer

    R = int(os.environ.get("TRAIN_RAYS", "4096"))
    S = int(os.environ.get("TRAIN_FIELD_SAMPLES", "48"))
    tcfg = PC.FruitNerfModelConfig(num_nerf_samples_per_ray=S)
    params = synthetic.p_rand(tcfg.field_spec(100), tcfg.proposal_specs(), seed=0, device=dev)
    model = FruitModel(tcfg, box, 100, {"semantics": Semantics()}, device=dev, params=params)
    model.training = True
    tr = FruitTrainer(model)
    g = torch.Generator().manual_seed(0)
    idx = torch.stack([torch.randint(0, 100, (R,), generator=g), torch.randint(0, H, (R,), generator=g),
                       torch.randint(0, W, (R,), generator=g)], -1)
    rb = cams.to(dev).generate_rays(idx.to(dev))
    batch = {"image": torch.rand(R, 3, generator=g).to(dev), "fruit_mask": (torch.rand(R, 1, generator=g) > 0.5).float().to(dev)}
    n_it = 6  # all of them update the proposal networks (the first ten iterations do)
    for _ in range(n_it):
        tr.train_iteration(rb, batch)
    units, unit = n_it, f"iterations of {R} rays x {S} field samples"
else:
    raise SystemExit(f"unknown workload {what!r}")
torch.cuda.synchronize()
print("PMC_UNITS " + json.dumps({"workload": what, "units": units, "unit": unit}), flush=True)

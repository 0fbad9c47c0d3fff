"""Throughput of the exporter / projection rows on cuda:0 at the reference's call shapes.
Profiling aid, not a test:  python tools/export_probe.py"""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import config as PC, ops, synthetic
from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig
from cropnerf_amd.fruit_nerf.export.exporter_utils import sample_volume
from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import generate_point_cloud
from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context
from cropnerf_amd.fruit_nerf.fruit_pipeline import FruitPipeline, FruitPipelineConfig
from cropnerf_amd.rays import Cameras, SceneBox

dev = "cuda"
NUM_POINTS = int(os.environ.get("NUM_POINTS", 1_000_000))  # BASELINE.json configs[3]: NUM_POINTS=10000000
H = W = 800
cfg = PC.FruitNerfModelConfig(matrix_precision=os.environ.get("MATRIX_PRECISION", "fp32"))  # or split_bf16
fspec = cfg.field_spec(100)
params = synthetic.p_rand(fspec, cfg.proposal_specs(), seed=0, device=dev)
params["field.mlp_base_mlp.layers.1.bias"][0] += 4.0       # some density / fruit so the exporters keep points
params["field.field_head_semantics.net.bias"] += 3.0
c2w, intr = synthetic.orbit_cameras(100, height=H, width=W)
cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], H, W)
box = SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
res = {}

def sync_time(fn, n=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, r

# (a16) dense volume export: 600 x 600 rays x 3000 samples in 512-ray calls (reference: 3000 x 3000 rays)
pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(4096, 512), cfg), dev, cams, box, test_mode="export", params=params)
pipe.model.setup_inference(True, 3000)
side = 600
n_rays = pipe.datamanager.setup_inference(((-1, -1, -1 + .318), (1, 1, 1 + .318)), side)
t, pcds = sync_time(lambda: sample_volume(pipe, n_rays, transform_json={"scale": 1.0, "transform": None}, capacity=1 << 26))
res["dense_export"] = {"rays": n_rays, "samples_per_ray": 3000, "rays_per_call": 512, "seconds": round(t, 3),
                       "field_samples_per_sec": n_rays * 3000 / t, "kept": {k: int(v["points"].shape[0]) for k, v in pcds.items()},
                       "full_3000x3000_estimate_s": round(t * (3000 * 3000) / n_rays, 1)}
# (a17) semantic point cloud: 2048-ray calls until 1e6 kept points
pipe2 = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(2048, 2048), cfg), dev, cams, box, test_mode="test", params=params)
t, pcd = sync_time(lambda: generate_point_cloud(pipe2, num_points=NUM_POINTS, remove_outliers=False))
t_sor, pcd_sor = sync_time(lambda: generate_point_cloud(pipe2, num_points=NUM_POINTS, remove_outliers=True))
res["pointcloud_export"] = {"kept_points": int(pcd["points"].shape[0]), "seconds": round(t, 3),
                            "train_batches": pipe2.datamanager.train_count, "rays_per_sec": pipe2.datamanager.train_count * 2048 / t}
res["pointcloud_export_with_outlier_removal"] = {"seconds": round(t_sor, 3), "kept_after_removal": int(pcd_sor["points"].shape[0])}
# (a15) projection: one (camera, sub-cluster AABB) job at 800x800 = AABB-restricted render + occlusion pass
m = pipe2.model
m.config.eval_num_rays_per_chunk = 4096
aabb = SceneBox(torch.tensor([[-0.15, -0.15, -0.15], [0.15, 0.15, 0.15]]))
with background_color_override_context(torch.zeros(3)):
    m.project_cluster(cams[0], aabb, 0)
    t, _ = sync_time(lambda: [m.project_cluster(cams[i], aabb, i) for i in range(10)])
res["projection"] = {"jobs": 10, "image": [H, W], "seconds_per_job": round(t / 10, 4), "chunk": 4096}
m.config.eval_num_rays_per_chunk = 1 << 15
with background_color_override_context(torch.zeros(3)):
    t, _ = sync_time(lambda: [m.project_cluster(cams[i], aabb, i) for i in range(10)])
res["projection_chunk32k"] = {"seconds_per_job": round(t / 10, 4)}
print(json.dumps(res))

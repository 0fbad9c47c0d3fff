"""Whole-image render of the default fruit_nerf_method in eval (proposal sampler (256, 96) + 48 field samples per ray,
fruit_nerf.py:377-404) at 800 x 800, at three values of eval_num_rays_per_chunk.  The model works in chunks of
max(eval_num_rays_per_chunk, FruitModel.EVAL_CHUNK = 2^18) rays; EVAL_CHUNK=<n> in the environment lowers that floor (13.95 /
13.22 / 12.68 ms per image in chunks of 2^15 / 2^16 / 2^18 rays is how the floor was chosen).  python tools/image_probe.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import config as PC, synthetic
from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
from cropnerf_amd.rays import Cameras, SceneBox

# IMPLEMENTATION=tcnn TABLE_DTYPE=float16 MATRIX_PRECISION=f16: a model as an imported reference checkpoint is (tcnn layout, fp16
# tables, random values) rendered in tcnn's own arithmetic class
impl = os.environ.get("IMPLEMENTATION", "torch")
cfg = PC.FruitNerfModelConfig(implementation=impl, hash_table_dtype=os.environ.get("TABLE_DTYPE", "float32"),
                              matrix_precision=os.environ.get("MATRIX_PRECISION", PC.FruitNerfModelConfig().matrix_precision))
params = synthetic.p_rand(cfg.field_spec(100), cfg.proposal_specs(), seed=0, device="cuda") if impl == "torch" else None
c2w, intr = synthetic.orbit_cameras(100)
cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], 800, 800).to("cuda")
res = {}
if os.environ.get("EVAL_CHUNK"):
    FruitModel.EVAL_CHUNK = FruitModel.JAGGED_CHUNK = int(os.environ["EVAL_CHUNK"])
for chunk in (1 << 15, 1 << 16, 1 << 18):
    cfg.eval_num_rays_per_chunk = chunk
    m = FruitModel(cfg, SceneBox(torch.tensor(synthetic.SCENE_AABB)), 100, {"semantics": Semantics()}, device="cuda",
                   test_mode="test", params=params)
    rb = cams.generate_rays(3, keep_shape=True)
    out = m.get_outputs_for_camera_ray_bundle(rb)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(5):
        out = m.get_outputs_for_camera_ray_bundle(cams.generate_rays(i, keep_shape=True))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    res[f"chunk_{chunk}"] = {"ms_per_image": round(dt * 1e3, 2), "images_per_sec": round(1 / dt, 1),
                             "rays_per_sec": 640000 / dt, "keys": sorted(out)}
print(json.dumps(res))

"""Throughput of the shape-generic path on the fruit_nerf_method_big field (geo 30, 3 x 128 semantic layers, 2^21-entry
levels, 128 samples per ray behind a (512, 256) proposal sampler).  Profiling aid:  python tools/big_shape_probe.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import synthetic
from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
from cropnerf_amd.fruit_nerf import fruit_nerf_config as FC
from cropnerf_amd.rays import Cameras, SceneBox

cfg = getattr(FC, os.environ.get("METHOD", "fruit_nerf_method_big")).config.pipeline.model
c2w, intr = synthetic.orbit_cameras(8, height=800, width=800)
cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], 800, 800).to("cuda")
m = FruitModel(cfg, SceneBox(torch.tensor(synthetic.SCENE_AABB)), 8, {"semantics": Semantics()}, device="cuda", test_mode="inference")
for k, v in m.params.items():
    if k.endswith("hash_table"):
        v.mul_(100.0)  # the 1e-3 init is an empty volume
PROBE = os.environ.get("PROBE", "both")  # render | train | both
rb = cams.generate_rays(0, keep_shape=False, aabb_box=SceneBox(torch.tensor(synthetic.SCENE_AABB)))
rb = rb[:65536] if hasattr(rb, "__getitem__") else rb
if PROBE != "train":
  out = m(rb)
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(3):
    out = m(rb)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t) / 3
  R = out["rgb"].shape[0]
  S = cfg.num_nerf_samples_per_ray
  print(json.dumps({"matrix_precision": m.config.matrix_precision, "rays": R, "field_samples_per_ray": S, "proposal_samples_per_ray": list(cfg.num_proposal_samples_per_ray),
                  "ms_per_call": round(dt * 1e3, 2), "rays_per_sec": R / dt, "field_samples_per_sec": R * S / dt,
                  "finite": bool(torch.isfinite(out["rgb"]).all())}))
if PROBE == "render":
    sys.exit(0)

# training iteration of the big method: 8192 rays (train_num_rays_per_batch = 4096 * 2, fruit_nerf_config.py)
from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
m.training = True
tr = FruitTrainer(m)
g = torch.Generator().manual_seed(0)
Rt = int(os.environ.get("TRAIN_RAYS", "8192"))
idx = torch.stack([torch.randint(0, 8, (Rt,), generator=g), torch.randint(0, 800, (Rt,), generator=g),
                   torch.randint(0, 800, (Rt,), generator=g)], -1)
rays = cams.generate_rays(idx.cuda())
batch = {"image": torch.rand(Rt, 3, generator=g).cuda(), "fruit_mask": (torch.rand(Rt, 1, generator=g) > 0.5).float().cuda()}
for _ in range(2):
    tr.train_iteration(rays, batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    tr.train_iteration(rays, batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(json.dumps({"train_rays": Rt, "ms_per_iter": round(dt * 1e3, 2), "rays_per_sec": Rt / dt}))

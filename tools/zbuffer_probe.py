"""Depth-based projection throughput: 10 M plant points as the occluder + 1 M fruit points, 1440 x 1920 buffers.
Profiling aid:  python tools/zbuffer_probe.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as M

rng = np.random.default_rng(0)
c2w = np.eye(4); c2w[:3, 3] = [0.0, 0.0, 2.5]
P = M.get_projection_mat(M.FX, M.FY, M.CX, M.CY, c2w)
tree = torch.as_tensor(rng.normal(size=(10_000_000, 3)) * 0.4).cuda()
fruit = torch.as_tensor(rng.normal(size=(1_000_000, 3)) * 0.1).cuda()
z = torch.full((M.IMG_H, M.IMG_W), float("inf"), dtype=torch.float32, device="cuda")
img = torch.zeros(M.IMG_H, M.IMG_W, dtype=torch.uint8, device="cuda")
def run():
    z.fill_(float("inf")); img.zero_()
    M.update_buffer(z, M.get_projection(P, tree), img, 0, large=True)
    return M.update_buffer(z, M.get_projection(P, fruit), img, 1)[2]
run(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): vis = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(json.dumps({"points": 11_000_000, "ms": round(dt * 1e3, 2), "points_per_sec": 11e6 / dt, "visible_pixels": int((vis > 0).sum())}))

"""Outlier pass and normals on a surface-like cloud of N points (thin spherical caps around an orbit of cameras: what the C4
export's kept points look like), timed; and against the exact search on a subsample.  python tools/knn_probe.py [N]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device="cuda").manual_seed(0)
cam = torch.randint(0, 100, (N,), device="cuda", generator=g)
ang = cam.float() * (2 * 3.14159265 / 100)
centre = torch.stack([0.8 * ang.cos(), 0.8 * ang.sin(), 0.2 * (cam % 2).float() - 0.1], -1)
d = torch.nn.functional.normalize(-centre + 0.35 * torch.randn(N, 3, device="cuda", generator=g), dim=-1)
pts = centre + d * (0.75 + 0.003 * torch.randn(N, 1, device="cuda", generator=g))
pts[:1000] = torch.rand(1000, 3, device="cuda", generator=g) * 4 - 2  # some isolated points far from everything
for name, fn in (("outlier mask (20 nn)", lambda: ops.statistical_outlier_mask(pts, 20, 10.0)), ("normals (30 nn)", lambda: ops.estimate_normals(pts, 30))):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name}: {N} points in {time.perf_counter() - t:.3f} s")
_, g, _, _ = ops._knn_grid(pts, 8.0)
print("top grid", list(g.top), "top cell", round(g.top_cell_size, 5), "sub", g.sub)
# exactness on a subsample: the mean neighbour distance against scipy
import numpy as np
from scipy.spatial import cKDTree
sub = pts[:: max(1, N // 200000)].contiguous()
got = ops.knn_mean_distance(sub, 20).cpu().numpy()
dist, _ = cKDTree(sub.cpu().numpy().astype(np.float64)).query(sub.cpu().numpy().astype(np.float64), k=20)
print("max |mean distance - exact|:", float(np.abs(got - dist.mean(1)).max()))

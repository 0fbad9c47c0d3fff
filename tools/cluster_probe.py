"""Super-cluster stage throughput: a synthetic semantic cloud of fruit-sized blobs + clutter through voxel down-sampling,
DBSCAN and statistical outlier removal (segmenter.get_super_clusters).  Profiling aid:  N=1000000 python tools/cluster_probe.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import ops

n = int(os.environ.get("N", 1_000_000)); vx = float(os.environ.get("VX", 1e-3)); blobs = int(os.environ.get("BLOBS", 300)); sigma = float(os.environ.get("SIGMA", 0.012))
g = torch.Generator(device="cuda").manual_seed(0)
centres = torch.rand(blobs, 3, device="cuda", generator=g) * 2 - 1
which = torch.randint(0, blobs, (n - n // 20,), device="cuda", generator=g)
pts = torch.cat([centres[which] + torch.randn(len(which), 3, device="cuda", generator=g) * sigma,
                 torch.rand(n // 20, 3, device="cuda", generator=g) * 2.4 - 1.2]).contiguous()
def timed(f, reps=3):
    out = f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): out = f()
    torch.cuda.synchronize(); return out, (time.perf_counter() - t) / reps * 1e3
(down, _), t_vox = timed(lambda: ops.voxel_down_sample(pts, vx))
(labels, core), t_db = timed(lambda: ops.dbscan(down, 20 * vx, 30))
kept = down[labels >= 0].contiguous()
_, t_sor = timed(lambda: ops.statistical_outlier_mask(kept, 20, 2.0))
(_, lab), t_all = timed(lambda: ops.get_super_clusters(pts, vx))
print(json.dumps({"points": n, "voxel": vx, "after_voxel": len(down), "clusters": int(labels.max()) + 1, "noise": int((labels < 0).sum()),
                  "voxel_ms": round(t_vox, 2), "dbscan_ms": round(t_db, 2), "outlier_ms": round(t_sor, 2), "pipeline_ms": round(t_all, 2),
                  "final_clusters": len(torch.unique(lab))}))

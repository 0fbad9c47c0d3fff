"""The C4 bench cloud (bench.py: export_pointcloud_c4) and the grid its k-nearest passes run on: points per occupied cell,
the fullest cells, and the time of both passes.  Profiling aid:  python tools/c4_cloud_stats.py [num_points]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cropnerf_amd import config as PC, ops, synthetic  # noqa: E402
from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig  # noqa: E402
from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import generate_point_cloud  # noqa: E402
from cropnerf_amd.fruit_nerf.fruit_pipeline import FruitPipeline, FruitPipelineConfig  # noqa: E402
from cropnerf_amd.rays import Cameras, SceneBox  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
cfg0, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
cfg = PC.FruitNerfModelConfig()
p2 = {k: v.clone() for k, v in params.items()}
p2["field.mlp_base_mlp.layers.1.bias"][0] += 4.0
p2["field.field_head_semantics.net.bias"] += 3.0 + float(os.environ.get("SEM_BIAS", "-0.9351"))
cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], bench.H, bench.W)
box = SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(2048, 2048), cfg), dev, cams, box, test_mode="test", params=p2)
st = {}
pcd = generate_point_cloud(pipe, num_points=N, remove_outliers=False, stats=st)
pts = torch.from_numpy(pcd["points"]).to(device=dev, dtype=torch.float32)
print("points", pts.shape[0], "of", st["rays"], "rays; extent", (pts.max(0).values - pts.min(0).values).tolist())
ps, g, order, (top_rank, cell_start) = ops._knn_grid(pts, 8.0)
cnt = (cell_start[1:] - cell_start[:-1]).long()
occ = cnt[cnt > 0]
print("top grid", list(g.top), "top cell", round(g.top_cell_size, 5), "sub", g.sub, "fine cell", round(g.top_cell_size / g.sub, 6),
      "occupied top cells", int((top_rank >= 0).sum()), "occupied fine cells", int(occ.numel()), "mean per occupied fine cell",
      float(occ.float().mean()), "max", int(occ.max()))
q = torch.quantile(occ.float()[: 10_000_000], torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
print("occupancy quantiles 50/90/99/99.9 %:", q.tolist())
# duplicates: points that coincide exactly
u = torch.unique(pts, dim=0).shape[0]
print("distinct points", u)
for name, fn in (("outlier mask", lambda: ops.statistical_outlier_mask(pts, 20, 10.0)), ("normals", lambda: ops.estimate_normals(pts, 30))):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(name, round(time.perf_counter() - t, 3), "s")

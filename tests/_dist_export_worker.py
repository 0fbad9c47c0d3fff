"""2-rank rehearsal (gloo, both ranks on cuda:0) of the sharded exporters: point-cloud export + projection jobs."""
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
from cropnerf_amd import config as PC, synthetic
from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig
from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import generate_point_cloud
from cropnerf_amd.fruit_nerf.fruit_nerf import Semantics, background_color_override_context
from cropnerf_amd.fruit_nerf.fruit_pipeline import FruitPipeline, FruitPipelineConfig
from cropnerf_amd.rays import Cameras, SceneBox
cfg = PC.FruitNerfModelConfig(log2_hashmap_size=14)
params = synthetic.p_rand(cfg.field_spec(8), cfg.proposal_specs(), seed=0, device="cuda")
params["field.mlp_base_mlp.layers.1.bias"][0] += 4.0
params["field.field_head_semantics.net.bias"] += 3.0
c2w, intr = synthetic.orbit_cameras(8, height=64, width=64, focal=90.0)
cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], 64, 64)
pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(512, 512), cfg), "cuda", cams, SceneBox(torch.tensor([[-1.0,-1,-1],[1,1,1]])),
                     test_mode="test", params=params, world_size=ws, local_rank=rank)
pcd = generate_point_cloud(pipe, num_points=6000, remove_outliers=False)
n = pcd["points"].shape[0]
first = pcd["points"][0].tolist()
# every rank holds the same gathered cloud; the two halves come from different random rays
allfirst = [None] * ws
dist.all_gather_object(allfirst, (n, first))
assert all(a == allfirst[0] for a in allfirst), allfirst
assert 6000 <= n < 6000 + 2 * 512 * 20, n
half = n // 2
assert not np.allclose(pcd["points"][:100], pcd["points"][half:half + 100])
class DS:
    cameras = cams
    metadata = {"semantics": Semantics()}
data = [{"aabb": np.array([[[-0.2, -0.2, -0.2], [0.2, 0.2, 0.2]], [[-0.1, 0.0, 0.0], [0.3, 0.3, 0.3]]]), "pcd": {}}]
with background_color_override_context(torch.zeros(3)), torch.no_grad():
    res = pipe.model.get_outputs_for_projections(DS, None, pcd_data=data, save=False)
keys = sorted(res)
allkeys = [None] * ws
dist.all_gather_object(allkeys, keys)
flat = sorted(k for ks in allkeys for k in ks)
assert len(flat) == 16 and len(set(flat)) == 16 and all(len(ks) == 8 for ks in allkeys), allkeys
# dense volume export: batches dealt round-robin + gather == the single-rank export (same point sets)
from unittest import mock
from cropnerf_amd.fruit_nerf.export.exporter_utils import sample_volume
pipe_e = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(512, 100), cfg), "cuda", cams, SceneBox(torch.tensor([[-1.0,-1,-1],[1,1,1]])),
                       test_mode="export", params=params, world_size=ws, local_rank=rank)
pipe_e.model.setup_inference(True, 50)
n_rays = pipe_e.datamanager.setup_inference(((-1, -1, -1), (1, 1, 1)), 21)
with torch.no_grad():
    sharded = sample_volume(pipe_e, n_rays, sem_thresh=0.5, den_thresh=5.0)
    pipe_e.datamanager.train_count = 0
    with mock.patch("cropnerf_amd.distributed.world", lambda group=None: (0, 1)):
        single = sample_volume(pipe_e, n_rays, sem_thresh=0.5, den_thresh=5.0)
for k in ("semantic_colormap", "semantic", "density"):
    a, b = sharded[k]["points"], single[k]["points"]
    assert a.shape == b.shape and a.shape[0] > 0, (k, a.shape, b.shape)
    assert np.allclose(a[np.lexsort(a.T)], b[np.lexsort(b.T)], atol=1e-6), k
if rank == 0: print("sharded export ok:", n, "points; projection jobs per rank", [len(k) for k in allkeys])
dist.barrier(); dist.destroy_process_group()

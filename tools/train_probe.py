"""Times the default fruit_nerf training iteration (TRAIN_RAYS random rays, default 4096) on cuda:0; CN_DEBUG_SKIP ablates parts of
cn_field_backward (1 hash atomics, 2 embedding atomics, 4 weight-gradient dots).  Profiling aid, not a test."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import config as PC, ops, synthetic
from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
from cropnerf_amd.rays import RayBundle, SceneBox, Cameras
dev = "cuda"
cfg = PC.FruitNerfModelConfig(log2_hashmap_size=int(os.environ.get("LOG2_T", "19")),
                              num_nerf_samples_per_ray=int(os.environ.get("TRAIN_FIELD_SAMPLES", "48")),
                              matrix_precision=os.environ.get("TRAIN_MATRIX", "fp32"))  # "f16": the mixed-precision iteration
fspec = cfg.field_spec(100)
params = synthetic.p_rand(fspec, cfg.proposal_specs(), seed=0, device=dev)
model = FruitModel(cfg, SceneBox(torch.tensor([[-1.0,-1,-1],[1,1,1]])), 100, {"semantics": Semantics()}, device=dev, params=params)
model.training = True
tr = FruitTrainer(model)
c2w, intr = synthetic.orbit_cameras(100)
cams = Cameras(c2w, intr[:,0], intr[:,1], intr[:,2], intr[:,3], 800, 800).to(dev)
g = torch.Generator().manual_seed(0)
R = int(os.environ.get("TRAIN_RAYS", "4096"))
idx = torch.stack([torch.randint(0,100,(R,),generator=g), torch.randint(0,800,(R,),generator=g), torch.randint(0,800,(R,),generator=g)],-1)
idx = idx.to(dev)
from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManager  # noqa: E402
_sort_default = "1" if idx.shape[0] >= FruitDataManager.SORT_BATCHES_FROM else "0"
if os.environ.get("SORT_BATCH", _sort_default) != "0":  # as FruitDataManager hands a batch out: sorted by camera and pixel from 16 384 rays on; SORT_BATCH=0 / 1 forces
    idx = idx[ops.ray_sort_permutation(idx, 800, 800)]
rb = cams.generate_rays(idx)
batch = {"image": torch.rand(R,3,generator=g).to(dev), "fruit_mask": (torch.rand(R,1,generator=g)>0.5).float().to(dev)}
WARM, ITERS = int(os.environ.get("WARM", "2")), int(os.environ.get("ITERS", "20"))
if os.environ.get("PER_ITER") == "1":  # one line per iteration, each waited for (which variant ran, and how long it took)
    for i in range(WARM + ITERS):
        torch.cuda.synchronize(); t = time.perf_counter()
        upd = tr.proposal_update_due(tr._sampler_step)
        tr.train_iteration(rb, batch)
        torch.cuda.synchronize(); print(f"it {i:3d} proposals_updated {int(upd)} ms {(time.perf_counter() - t) * 1e3:7.3f}")
    sys.exit(0)
for i in range(WARM): tr.train_iteration(rb, batch)
torch.cuda.synchronize(); t=time.perf_counter()
for i in range(ITERS): tr.train_iteration(rb, batch)
torch.cuda.synchronize(); print("CN_DEBUG_SKIP", os.environ.get("CN_DEBUG_SKIP"), "rays", R, "ms/iter", (time.perf_counter()-t)/ITERS*1e3)

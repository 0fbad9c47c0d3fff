"""Effect of cn_render_opts.early_stop_transmittance on an opaque scene (the synthetic P-rand scene is translucent, so
bench.py never stops a ray).  Profiling aid, not a test:  python tools/early_stop_probe.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import config as PC, ops, synthetic

dev = "cuda"
H = W = 800
S, R = 192, 65536
cfg = PC.FruitNerfModelConfig()
fspec = cfg.field_spec(100)
c2w, intr = synthetic.orbit_cameras(100, height=H, width=W)
c2w, intr = c2w.to(dev), intr.to(dev)
sc = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), True)
res = {}
for boost in [float(b) for b in os.environ.get("BOOSTS", "0,3,5").split(",")]:
    params = synthetic.p_rand(fspec, cfg.proposal_specs(), seed=0, device=dev)
    params["field.mlp_base_mlp.layers.1.bias"][0] += boost
    fh = ops.FieldHandle(params, fspec)
    rg = ops.raygen_pinhole(c2w, intr, cam=0, height=H, width=W, pixel_start=200 * W, num_rays=R)
    n, f = ops.intersect_aabb(rg["origins"], rg["directions"], [-1.0, -1, -1, 1, 1, 1])
    row = {}
    for eps in (0.0, 1e-4, 1e-2):
        opts = ops.render_opts(S, image_width=W, pixel_start=200 * W, early_stop_transmittance=eps)
        out = ops.render_rays(fh, sc, opts, rg["origins"], rg["directions"], n, f)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            out = ops.render_rays(fh, sc, opts, rg["origins"], rg["directions"], n, f)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 30 * 1e3
        if eps == 0.0:
            ref = out
        row[f"eps={eps:g}"] = {"ms": round(ms, 3), "max_rgb_err": float((out["rgb"] - ref["rgb"]).abs().max()),
                               "mean_acc": float(out["accumulation"].mean())}
    res[f"density_logit+{boost:g}"] = row
print(json.dumps(res, indent=1))

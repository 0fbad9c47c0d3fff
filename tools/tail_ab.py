"""A/B of the packed schedule of 48-sample rays (render_split_kernel<..., PACK>: two rays in three half-steps):
CN_SPLIT_PACK=0 keeps one ray per two half-steps.  Times the composited render (eval) and the per-sample render (training
forward) of 65 536-ray batches at S = 48 and checks that the two schedules give the same bits (odd ray counts, with and without
the image hint, weights requested, S = 33 .. 48).       python tools/tail_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cropnerf_amd import ops, _lib as L

dev = torch.device("cuda", 0)
cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
batches = bench.make_batches(ops, c2w, intr, 0, 1)
scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=True)
os.environ["CN_FUSED_SPLIT"] = "2"  # the producer/consumer kernel also for the small checks


def run(kind, S, b, n=None, hint=True, weights=False, bins=None):
    o, d, ne, f, cam, start = b
    if n is not None:
        o, d, ne, f = (t[:n].contiguous() for t in (o, d, ne, f))
        bins = None if bins is None else bins[:n].contiguous()
    if kind == "rays":
        opts = ops.render_opts(S, image_width=800 if hint else 0, pixel_start=start if hint else 0)
        return ops.render_rays(fh, scene, opts, o, d, ne, f, bins=bins, want_weights=weights)
    return ops.render_samples(fh, scene, ops.render_opts(S), o, d, ne, f, bins=bins)


bins48 = [ops.proposal_sample(dh, scene, *b[:4], cfg.num_proposal_samples_per_ray, 48)["euclidean_bins"] for b in batches]
for kind in ("rays", "samples"):
    res = {}
    for pack in ("0", "1"):
        os.environ["CN_SPLIT_PACK"] = pack
        res[pack] = bench.launch_stats(lambda i: run(kind, 48, batches[i % len(batches)], bins=bins48[i % len(batches)]), n=40, warm=5)
    print(f"{kind:8s} 65536 rays x 48 samples: one ray per two half-steps {res['0']['median'] * 1e3:.3f} ms   "
          f"packed {res['1']['median'] * 1e3:.3f} ms", flush=True)
bad = 0
for kind in ("rays", "samples"):
    for S in (48, 47, 40, 33):
        for n in (65536, 65535, 4099, 1001, 9, 1):
            for hint in (True, False):
                for use_bins in (True, False):
                    if kind == "samples" and hint:
                        continue
                    bn = bins48[1][:, : S + 1].contiguous() if use_bins else None
                    outs = {}
                    for pack in ("0", "1"):
                        os.environ["CN_SPLIT_PACK"] = pack
                        outs[pack] = run(kind, S, batches[1], n=n, hint=hint, weights=(kind == "rays"), bins=bn)
                    torch.cuda.synchronize()
                    same = all(torch.equal(outs["0"][k], outs["1"][k]) for k in outs["0"])
                    if not same:
                        bad += 1
                        worst = max(float((outs["0"][k].float() - outs["1"][k].float()).abs().max()) for k in outs["0"])
                        print(f"MISMATCH {kind} S={S} n={n} hint={hint} bins={use_bins}: max |diff| {worst:.3e}", flush=True)
print("bit-identity checks:", "all equal" if bad == 0 else f"{bad} mismatches")

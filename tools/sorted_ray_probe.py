"""Would the point-cloud exporter's random rays render faster if a LAUNCH of many calls were sorted by camera and pixel first?
N random (camera, row, col) draws, rendered in 65 536-ray chunks (proposal sampler + 48 field samples, eval) in draw order and in
(camera, Morton(row, col)) order.  Profiling aid:  python tools/sorted_ray_probe.py [N]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cropnerf_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
dev = torch.device("cuda", 0)
cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=True)
S, CH = 48, 65536
g = torch.Generator().manual_seed(0)
idx = torch.floor(torch.rand(N, 3, generator=g) * torch.tensor([c2w.shape[0], bench.H, bench.W])).long().to(dev)


def morton(r, c):
    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return spread(r) | (spread(c) << 1)


def run(order_name, ids):
    ts_s, ts_f = [], []
    for k in range(0, max(min(ids.shape[0], 16 * CH) - CH + 1, 1), CH):
        sub = ids[k:k + CH].contiguous()
        CHn = sub.shape[0]
        r = ops.raygen_pinhole(c2w, intr, ray_indices=sub)
        o, d = r["origins"], r["directions"]
        n = torch.full((CHn, 1), 0.05, device=dev)
        f = torch.full((CHn, 1), 1000.0, device=dev)
        for rep in range(3):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            ps = ops.proposal_sample(dh, scene, o, d, n, f, cfg.num_proposal_samples_per_ray, S)
            e[1].record()
            ops.render_rays(fh, scene, ops.render_opts(S), o, d, n, f, bins=ps["euclidean_bins"])
            e[2].record()
            e[2].synchronize()
        ts_s.append(e[0].elapsed_time(e[1]))
        ts_f.append(e[1].elapsed_time(e[2]))
    print(f"{order_name:28s}: sampler {statistics.median(ts_s):.3f} ms, field {statistics.median(ts_f):.3f} ms per {CH} rays")


run("draw order", idx)
key = idx[:, 0] * (1 << 22) + morton(idx[:, 1], idx[:, 2])
run(f"sorted, launch of {N} rays", idx[torch.argsort(key)])
for n_small in (1 << 20, 1 << 18, 1 << 16):
    sub = idx[:n_small]
    k2 = sub[:, 0] * (1 << 22) + morton(sub[:, 1], sub[:, 2])
    run(f"sorted, launch of {n_small} rays", sub[torch.argsort(k2)])

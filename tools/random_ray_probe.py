"""Random against coherent rays through the eval forward of the default method (proposal sampler + 48 field samples), by render
kernel form (CN_FUSED_SPLIT=0: one wave per ray; 1: the producer/consumer kernel) -- the forward the point-cloud exporter and
the training iteration pay.  Profiling aid:  python tools/random_ray_probe.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cropnerf_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=True)
R, S = 65536, int(os.environ.get("S", "48"))
g = torch.Generator().manual_seed(0)


def rays(random_pixels: bool):
    if random_pixels:
        idx = torch.floor(torch.rand(R, 3, generator=g) * torch.tensor([c2w.shape[0], bench.H, bench.W])).long().to(dev)
        r = ops.raygen_pinhole(c2w, intr, ray_indices=idx)
    else:
        r = ops.raygen_pinhole(c2w, intr, cam=3, height=bench.H, width=bench.W, pixel_start=200 * bench.W, num_rays=R)
    n = torch.full((R, 1), 0.05, device=dev)
    f = torch.full((R, 1), 1000.0, device=dev)
    return r["origins"], r["directions"], n, f


def timed(fn, n=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


for kind in ("coherent", "random"):
    o, d, n, f = rays(kind == "random")
    ps = ops.proposal_sample(dh, scene, o, d, n, f, cfg.num_proposal_samples_per_ray, S)
    t_s = timed(lambda: ops.proposal_sample(dh, scene, o, d, n, f, cfg.num_proposal_samples_per_ray, S))
    for split in ("0", "1", "2"):
        os.environ["CN_FUSED_SPLIT"] = split
        t_r = timed(lambda: ops.render_rays(fh, scene, ops.render_opts(S), o, d, n, f, bins=ps["euclidean_bins"]))
        t_p = timed(lambda: ops.render_samples(fh, scene, ops.render_opts(S), o, d, n, f, bins=ps["euclidean_bins"]))
        print(f"{kind:9s} rays, S={S}: sampler {t_s:.3f} ms; CN_FUSED_SPLIT={split}: render_rays {t_r:.3f} ms, render_samples {t_p:.3f} ms")
    del os.environ["CN_FUSED_SPLIT"]

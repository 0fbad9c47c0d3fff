#!/usr/bin/env python3
"""bench.py -- throughput of the fruit_nerf volumetric ray-marching hot path on MI355X.

    python bench.py --gpus 1 --steps 40 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8(d) mode M-uniform): synthetic plant_1-shaped scene (parameter set
P-rand, full-size field: 16-level 2^19-entry hash grid + the three tiny MLPs), 800x800 pinhole cameras on an orbit,
192 field samples per ray between the ray's entry and exit of the +-1 scene box, 65 536-ray batches.
One step = one batch through the fused hot path (sampler + field + compositor, cn_render_rays) with the rays
already resident in HBM.  With N > 1 GPUs every rank renders its own batch (rays sharded by batch: weak scaling)
and the per-ray outputs (rgb, accumulation, depth, semantics: 24 B/ray) are all-gathered over RCCL inside the
timed step -- the only exchange the sharded path has.

Prints ONE JSON line (rank 0).  `value` = field samples per second over all GPUs.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

H = W = 800
FOCAL = 1111.1
S = 192
R = 65536
NUM_CAMERAS = 100
DISTINCT_BATCHES = 10  # ~ one 800x800 view (640 000 rays) worth of different batches, cycled
BYTES_PER_SAMPLE = 16 * 8 * 2 * 4  # 16 levels x 8 corners x 2 features x fp32 = 1024 B of table reads
BYTES_PER_RAY_IO = (3 + 3 + 1 + 1) * 4 + (3 + 1 + 1 + 1 + 3) * 4  # o,d,near,far in; rgb,acc,depth,sem,cmap out
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["uniform", "proposal"], default="uniform",
                    help="uniform: 192 field evals/ray (headline); proposal: + (256,96) proposal-net evals/ray")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-image-hint", action="store_true",
                    help="do not tell the renderer that a batch is a pixel run of an 800-wide image (A/B of the XCD stripe mapping)")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step timing")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline workload only (use under rocprofv3 so per-kernel averages are not mixed with the "
                         "proposal-mode and training launches)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    return ap.parse_args()


def setup_dist(n_gpus: int):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal hook for a one-GPU box: BENCH_REHEARSE_ON_ONE_GPU=1 puts every rank on cuda:0 and exchanges over
        # gloo (host staging).  Never set by the driver; the real path below is one rank per GPU over RCCL.
        if os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {n_gpus}")
    return rank, local, world


def build_scene(device):
    from cropnerf_amd import config, ops, synthetic

    cfg = config.FruitNerfModelConfig(num_nerf_samples_per_ray=S)
    fspec = cfg.field_spec(num_images=NUM_CAMERAS)
    pspecs = cfg.proposal_specs()
    params = synthetic.p_rand(fspec, pspecs, seed=0, device=device)
    fh = ops.FieldHandle(params, fspec)
    dh = [ops.DensityHandle(params, i, ps) for i, ps in enumerate(pspecs)]
    c2w, intr = synthetic.orbit_cameras(NUM_CAMERAS, height=H, width=W, focal=FOCAL)
    return cfg, fspec, pspecs, params, fh, dh, c2w.to(device), intr.to(device)


def make_batches(ops, c2w, intr, rank: int, world: int):
    """DISTINCT_BATCHES ray batches per rank, resident in HBM.  Batch b of rank r = R consecutive pixels of a camera
    that no other rank uses (ray batches are the sharding unit)."""
    batches = []
    aabb6 = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    for b in range(DISTINCT_BATCHES):
        cam = (rank + world * b) % NUM_CAMERAS
        start = (b * R) % (H * W - R)
        rays = ops.raygen_pinhole(c2w, intr, cam=cam, height=H, width=W, pixel_start=start, num_rays=R)
        nears, fars = ops.intersect_aabb(rays["origins"], rays["directions"], aabb6)
        batches.append((rays["origins"], rays["directions"], nears, fars, rays["camera_indices"][:, 0].contiguous(), start))
    return batches


def main():
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    rank, local, world = setup_dist(args.gpus)
    device = torch.device("cuda", local)
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops

    cfg, fspec, pspecs, params, fh, dh, c2w, intr = build_scene(device)
    batches = make_batches(ops, c2w, intr, rank, world)
    scene_u = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)
    scene_c = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=True)
    def opts_for(start):
        return ops.render_opts(S) if args.no_image_hint else ops.render_opts(S, image_width=W, pixel_start=start)

    opts = ops.render_opts(S)
    gather_buf = None
    if world > 1:
        import torch.distributed as dist

        # concatenated along dim 0 (valid for RCCL and gloo); two buffers: the all-gather of step i runs on RCCL's stream
        # under the render of step i+1
        gather_bufs = [torch.empty(world * R, 6, device=device) for _ in range(2)]
        gather_buf = gather_bufs[0]
    pending = []

    def step(i: int):
        o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
        if args.mode == "uniform":
            out = ops.render_rays(fh, scene_u, opts_for(start), o, d, n, f)
        else:
            ps = ops.proposal_sample(dh, scene_c, o, d, n, f, cfg.num_proposal_samples_per_ray, S)
            out = ops.render_rays(fh, scene_c, opts_for(start), o, d, n, f, bins=ps["euclidean_bins"])
        if world > 1:
            packed = torch.cat([out["rgb"], out["accumulation"], out["depth"], out["semantics"]], dim=-1)
            if dist.get_backend() == "gloo":  # rehearsal only
                host = torch.empty(gather_buf.shape)
                dist.all_gather_into_tensor(host, packed.cpu())
                gather_buf.copy_(host)
            else:
                if len(pending) == 2:
                    pending.pop(0).wait()  # stream-side wait: the buffer about to be reused has been filled
                pending.append(dist.all_gather_into_tensor(gather_bufs[i % 2], packed, async_op=True))
        return out

    def barrier():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (render_fused_kernel), measured live with HIP events on the launch stream
    roofline = None
    extra = {}
    if rank == 0:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        durs = []
        for i in range(min(args.steps, 20)):
            o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
            ev0.record()
            ops.render_rays(fh, scene_u, opts_for(start), o, d, n, f)
            ev1.record()
            ev1.synchronize()
            durs.append(ev0.elapsed_time(ev1) * 1e-3)
        avg = sum(durs) / len(durs)
        alg_bytes = R * (S * BYTES_PER_SAMPLE + BYTES_PER_RAY_IO)
        achieved = alg_bytes / avg / 1e9
        traffic, traffic_note = pmc_traffic()
        kernel_name = ("render_fused_kernel<false,false>" if os.environ.get("CN_FUSED_SPLIT", "1") == "0"
                       else "render_split_kernel")
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": traffic, "traffic_note": traffic_note, "algorithmic_bytes_per_launch": alg_bytes,
                    "avg_launch_ms": round(avg * 1e3, 4),
                    # second bound, reported beside the first: the folded MLP (9216 MAC per sample) on the fp32 matrix
                    # pipe (v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD = 157.3 TFLOP/s dense at 2.4 GHz)
                    "mlp_tflops_fp32": round(R * S * 2 * 9216 / avg / 1e12, 2), "mfma_peak_tflops_fp32": 157.3,
                    "mfma_frac": round(R * S * 2 * 9216 / avg / 1e12 / 157.3, 4)}
        extra["uniform_samples_per_sec_single_launch"] = R * S / avg

    # ---- CPU baseline (rank 0, N=1): the oracle ("port") on a bounded sample of the same workload ---------------
    cpu_baseline = None
    psnr = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, psnr = run_cpu_baseline(args, params, fspec, pspecs, batches, ops, fh, scene_u, opts)

    # ---- secondary numbers (rank 0, N=1): proposal-mode render and training iterations --------------------------------
    if rank == 0 and world == 1 and not args.no_secondary:
        extra.update(secondary_timings(args, cfg, params, fspec, batches, ops, fh, dh, scene_c, opts))

    if rank == 0:
        samples = world * R * S * args.steps
        line = {
            "metric": "ray_samples_per_sec",
            "value": samples / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "plant_1-shaped synthetic scene (P-rand), 800x800, 192 samples/ray, 65536-ray batch"
                                   f", mode M-{args.mode}, fused cn_render_rays, eval (no jitter)",
                       "rays_per_batch": R, "samples_per_ray": S, "image": [H, W], "mode": args.mode,
                       "sharding": "ray batches per rank + RCCL all-gather of per-ray outputs" if world > 1 else "none"},
            "rays_per_sec": world * R * args.steps / elapsed,
            "psnr_vs_oracle_db": psnr,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic():
    """Memory-side bytes per launch of render_fused_kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc_render_fused.json: FETCH_SIZE and WRITE_SIZE, KiB, separate passes).  No 2x FETCH_SIZE
    correction is applied: the guide calibrates that factor for 16-B-per-lane streaming reads only, these are 8-B
    gathers; the TCC miss count (x64 B) of the same run agrees with the raw value.  Infinity-Cache hits are included,
    so this is an upper bound on HBM bytes (the 64 MB table is cache-resident)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_render_fused.json")))
    if not files:
        return None, "no PMC summary under profiles/"
    try:
        with open(files[-1]) as fh_:
            d = json.load(fh_)
        val = lambda k: d[k]["avg_per_launch"] if isinstance(d[k], dict) else d[k]
        return int((val("FETCH_SIZE") + val("WRITE_SIZE")) * 1024), f"from {os.path.basename(files[-1])} (measured on the kernel version named there)"
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable PMC summary: {e}"


def secondary_timings(args, cfg, params, fspec, batches, ops, fh, dh, scene_c, opts):
    """Not the headline: M-proposal render (256+96 proposal-net evals + 192 field evals per ray) and the training
    iteration of the default method config (4096 rays: (256, 96) proposal + 48 field samples, forward + backward + Adam)."""
    out = {}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, n):
        fn(0)
        torch.cuda.synchronize()
        ev0.record()
        for i in range(n):
            fn(i)
        ev1.record()
        ev1.synchronize()
        return ev0.elapsed_time(ev1) * 1e-3 / n

    def prop(i):
        o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
        ps = ops.proposal_sample(dh, scene_c, o, d, n, f, cfg.num_proposal_samples_per_ray, S)
        o_ = opts if args.no_image_hint else ops.render_opts(S, image_width=W, pixel_start=start)
        ops.render_rays(fh, scene_c, o_, o, d, n, f, bins=ps["euclidean_bins"])

    # the headline workload with the optional split-bf16 matrix products (cn_render_opts.matrix_precision = 1; NOT the
    # headline, which is exact fp32): same batches, same kernel, same parity bar against the oracle (tests)
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops as _ops

    scene_u = _ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)

    def split_bf16(i):
        o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
        o_ = ops.render_opts(S, matrix_precision=L.MATRIX_SPLIT_BF16, **({} if args.no_image_hint else
                                                                           {"image_width": W, "pixel_start": start}))
        return ops.render_rays(fh, scene_u, o_, o, d, n, f)

    t = timed(split_bf16, 20)
    o, d, n, f, cam, start = batches[0]
    exact = ops.render_rays(fh, scene_u, ops.render_opts(S, image_width=W, pixel_start=start), o, d, n, f)["rgb"]
    mse = float(((split_bf16(0)["rgb"] - exact) ** 2).mean())
    out["uniform_mode_split_bf16_matrix"] = {
        "ms_per_batch": round(t * 1e3, 3), "samples_per_sec": R * S / t, "rays_per_sec": R / t,
        "psnr_vs_fp32_render_db": round(-10.0 * math.log10(max(mse, 1e-30)), 1),
        "note": "optional arithmetic (operands split into bf16 hi + lo, fp32 accumulation); the headline value is exact fp32"}

    t = timed(prop, 10)
    out["proposal_mode"] = {"ms_per_batch": round(t * 1e3, 3), "rays_per_sec": R / t, "field_samples_per_sec": R * S / t,
                            "network_evals_per_sec": R * (S + sum(cfg.num_proposal_samples_per_ray)) / t}
    if not args.no_train:
        from cropnerf_amd import config as PC
        from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
        from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
        from cropnerf_amd.rays import RayBundle, SceneBox

        tcfg = PC.FruitNerfModelConfig()  # reference defaults: 48 field samples, (256, 96) proposal samples
        model = FruitModel(tcfg, SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), NUM_CAMERAS,
                           {"semantics": Semantics()}, device=batches[0][0].device, params=params)
        model.training = True
        tr = FruitTrainer(model)
        g = torch.Generator().manual_seed(0)
        train = {}
        from cropnerf_amd import synthetic
        from cropnerf_amd.rays import Cameras

        c2w, intr = synthetic.orbit_cameras(NUM_CAMERAS, height=H, width=W, focal=FOCAL)
        cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], H, W).to(batches[0][0].device)
        for nrays in (4096, 65536):
            # training batches are random pixels over all images (PixelSampler), not a block of one image
            idx = torch.stack([torch.randint(0, NUM_CAMERAS, (nrays,), generator=g), torch.randint(0, H, (nrays,), generator=g),
                               torch.randint(0, W, (nrays,), generator=g)], -1)
            rb = cams.generate_rays(idx.to(batches[0][0].device))
            batch = {"image": torch.rand(nrays, 3, generator=g), "fruit_mask": (torch.rand(nrays, 1, generator=g) > 0.5).float()}
            batch = {k: v.to(batches[0][0].device) for k, v in batch.items()}
            t = timed(lambda i: tr.train_iteration(rb, batch), 5)
            train[str(nrays)] = {"ms_per_iter": round(t * 1e3, 3), "rays_per_sec": nrays / t}
        out["train_iteration"] = train
    return out


def run_cpu_baseline(args, params, fspec, pspecs, batches, ops, fh, scene_u, opts):
    """Time the CPU oracle (kind "port": this repo's op-for-op PyTorch restatement; the reference itself needs
    nerfstudio and cannot run) on a bounded sample of batch 0, all host cores; also PSNR of the GPU render vs it."""
    from oracle import field as OF
    from oracle import model as OM
    from oracle import rays as ORY

    # the GPU box hands one GPU a 16-core share of the host; os.cpu_count() reports the whole machine
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    cpu_params = {k: v.detach().cpu() for k, v in params.items()}
    g = fspec.grid
    ofs = OF.FieldSpec(grid=OF.GridSpec(g.num_levels, g.min_res, g.max_res, g.log2_hashmap_size), num_images=fspec.num_images)
    ops_ = [OF.ProposalSpec(OF.GridSpec(p.grid.num_levels, p.grid.min_res, p.grid.max_res, p.grid.log2_hashmap_size))
            for p in pspecs]
    model = OM.OracleModel(cpu_params, OM.ModelConfig(field=ofs, proposals=ops_, disable_scene_contraction=True),
                           torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), test_mode="inference")
    model.uniform_samples = S
    o, d, n, f, cam = (t.detach().cpu() for t in batches[0][:5])
    chunk = 1024
    idx = torch.arange(0, R, R // chunk)[:chunk]  # spread over the batch

    def bundle(sel):
        return ORY.RayBundle(o[sel], d[sel], torch.zeros(len(sel), 1), None, n[sel], f[sel])

    with torch.no_grad():
        ref = model.forward(bundle(idx))  # warm-up + PSNR reference
        t0 = time.perf_counter()
        done = 0
        k = 0
        while time.perf_counter() - t0 < args.cpu_baseline_seconds:
            sel = (idx + 1 + k) % R
            model.forward(bundle(sel))
            done += chunk
            k += 1
        dt = time.perf_counter() - t0
    gpu = ops.render_rays(fh, scene_u, opts, *(t[idx.to(t.device)].contiguous() for t in batches[0][:4]))
    mse = torch.mean((gpu["rgb"].cpu() - ref["rgb"]) ** 2).item()
    psnr = 10.0 * math.log10(1.0 / max(mse, 1e-20))
    base = {"value": done * S / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{done} rays x {S} samples of batch 0 in {chunk}-ray chunks, {dt:.1f} s, torch {torch.get_num_threads()} threads"}
    return base, round(psnr, 2)


if __name__ == "__main__":
    main()

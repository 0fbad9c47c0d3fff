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
BYTES_PER_SAMPLE_F16 = 16 * 8 * 2 * 2  # the same gathers from a half table (tcnn's parameter type): 512 B
BYTES_PER_PROPOSAL_SAMPLE = 5 * 8 * 2 * 4  # 5 levels x 8 corners x 2 features x fp32 = 320 B
BYTES_PER_RAY_IO = (3 + 3 + 1 + 1) * 4 + (3 + 1 + 1 + 1 + 3) * 4  # o,d,near,far in; rgb,acc,depth,sem,cmap out
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_FP32_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz (dense)
MLP_MAC_PER_SAMPLE = 9216  # the folded field MLPs as the render kernels evaluate them
L2_GATHER_PEAK_GBPS = 16800.0  # MI355X_MICROARCH.md, "Indexed rows": rows served from the XCD's L2, chip-wide (lower figure)
MFMA_F16_PEAK_TFLOPS = 2516.6  # v_mfma_f32_16x16x32_f16: 1024 FLOP/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz (dense; ~2.5 PF)
L1_LOOKUPS_PER_SEC = 256 * 2.4e9  # per-lane-addressed loads: one L1 line lookup per clock and CU (tools/gather_rate_microbench.hip)
ATOMIC_REQUESTS_PER_SEC = 21.07e9  # float-atomic requests the memory side takes (tools/atomic_microbench.hip)
# float-atomic requests per iteration, MEASURED with the TCC atomic counters (TCC_ATOMIC == TCC_EA0_ATOMIC) on the default method at
# every (rays, field samples per ray) bench.py times -- the set of cell-major levels depends on the batch size -- read from the
# newest profiles/r*_pmc_train_atomics.json; nothing is scaled or estimated any more (round 3 scaled the 192-sample figure).


def _measured_atomic_requests():
    """({"<rays>x<samples>": requests per iteration}, source).  The timed iterations all update both proposal networks (one field
    backward, two proposal backwards, their fold kernels): so do the profiled ones (tools/pmc_workloads.py train)."""
    import glob
    import json as _json

    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_train_atomics.json")), reverse=True):
        try:
            d = _json.load(open(f))
            if "configs" in d:  # round 4 format: per iteration, every kernel of the process
                got = {k: float(v["requests_per_iteration"]) for k, v in d["configs"].items()}
            else:  # rounds 1-3: per launch of the named kernels at 48 field samples per ray
                got = {}
                for rays, k in d["rays"].items():
                    avg = lambda part: next((v["TCC_ATOMIC_sum"]["avg_per_launch"] for n, v in k.items() if part in n), 0.0)
                    got[f"{rays}x48"] = avg("field_backward_mfma_kernel") + 2 * avg("proposal_backward_kernel") + \
                        3 * (avg("cell_scatter_fold_kernel") + avg("coarse_scatter_reduce_kernel"))
            if got:
                return got, f"profiles/{os.path.basename(f)} @ {d.get('commit', '?')} (TCC_ATOMIC per iteration at each batch size, measured)"
        except (KeyError, StopIteration, ValueError, OSError):
            continue
    return {}, "no profiles/r*_pmc_train_atomics.json"


ATOMIC_REQUESTS_PER_ITERATION, ATOMIC_REQUESTS_SOURCE = _measured_atomic_requests()


def workload_traffic(name: str, units: float):
    """(memory-side bytes for `units` of a secondary workload, source) from the newest profiles/r*_pmc_workloads.json
    (tools/collect_pmc_workloads.sh: FETCH_SIZE + WRITE_SIZE of this library's kernels over a whole run of the workload, separate
    rocprofv3 --pmc passes, divided by the units it processed).  Counters cannot be read inside this process: the figure is the
    profiled bytes per unit times the units of THIS run, labelled with its source.  Raw counter values (no x2 on FETCH_SIZE: the
    guide calibrates that for 16-byte streaming reads; these are 8-byte gathers and 4-byte atomics); Infinity-Cache hits count."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_workloads.json")), reverse=True):
        try:
            d = json.load(open(f))
            e = d["workloads"][name]
            return int(e["bytes_per_unit"] * units), f"profiles/{os.path.basename(f)} @ {d.get('commit', '?')}: {e['bytes_per_unit']:.1f} B per {e['unit']}"
        except (KeyError, ValueError, OSError):
            continue
    return None, "no profiles/r*_pmc_workloads.json entry"



def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["uniform", "proposal"], default="uniform",
                    help="uniform: 192 field evals/ray (headline); proposal: + (256,96) proposal-net evals/ray")
    ap.add_argument("--variant", choices=["headline", "exact_fp32", "tcnn_f16", "tcnn_f16_mfma", "split_bf16"], default="headline",
                    help="profiling aid (rocprofv3 -- python3 bench.py --variant ... --no-secondary --no-cpu-baseline): the "
                         "timed loop renders the M-uniform workload through another table / matrix mode; the JSON line "
                         "then names the variant and is NOT the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-image-hint", action="store_true",
                    help="do not tell the renderer that a batch is a pixel run of an 800-wide image (A/B of the XCD stripe mapping)")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step timing")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline workload only (use under rocprofv3 so per-kernel averages are not mixed with the "
                         "proposal-mode and training launches)")
    ap.add_argument("--no-subsystems", action="store_true",
                    help="skip the exporter / projection lines (10 M-point export, dense export, projection job)")
    ap.add_argument("--cpu-baseline-chunks", type=int, default=10,
                    help="timed chunks of the CPU baseline (after 3 warm-ups; the median chunk time is reported)")
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


def self_launch(args) -> int:
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes under
    torch.distributed.run (exactly the command the driver documents), before this process has made any GPU call, relay
    their output (rank 0 prints the one JSON line) and return the launcher's exit code.  Never exec: a process that has
    touched the GPU must not be replaced, and this one stays a plain parent that has not touched it."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: WORLD_SIZE is unset and --gpus {args.gpus} > 1: launching {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))  # before torch.cuda is touched in this process
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
    # The headline runs in the product's default arithmetic (FruitNerfModelConfig.matrix_precision, "split_bf16" since round 5:
    # bf16 hi + lo operands on the bf16 matrix pipe, fp32 accumulation -- the same parity bars against the fp32 oracle as the
    # exact kernels, tests/test_gpu_parity.py); the exact-fp32 products of rounds 1-4 are timed as `uniform_mode_exact_fp32`.
    from cropnerf_amd import config as _PC

    mp_name = _PC.FruitNerfModelConfig().matrix_precision
    mp = {"fp32": L.MATRIX_FP32, "split_bf16": L.MATRIX_SPLIT_BF16, "f16": L.MATRIX_F16}[mp_name]

    def opts_for(start):
        return (ops.render_opts(S, matrix_precision=mp) if args.no_image_hint
                else ops.render_opts(S, image_width=W, pixel_start=start, matrix_precision=mp))

    opts = ops.render_opts(S, matrix_precision=mp)
    fh_run, variant_kw = fh, {}
    if args.variant != "headline":
        fh_run, variant_kw = variant_field(args.variant, params, fspec, fh, device)
    gather_buf = None
    if world > 1:
        import torch.distributed as dist

        # concatenated along dim 0 (valid for RCCL and gloo); two buffers: the all-gather of step i runs on RCCL's stream
        # under the render of step i+1
        gather_bufs = [torch.empty(world * R, 6, device=device) for _ in range(2)]
        gather_buf = gather_bufs[0]
    pending = []
    last = {}  # the last step's packed outputs and the buffer they were gathered into (checked after the timed region)

    def step(i: int):
        o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
        if args.mode == "uniform" and args.variant != "headline":
            out = ops.render_rays(fh_run, scene_u, ops.render_opts(S, image_width=W, pixel_start=start, **variant_kw), o, d, n, f)
        elif args.mode == "uniform":
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
                last["packed"], last["buf"] = packed, gather_buf
            else:
                if len(pending) == 2:
                    pending.pop(0).wait()  # stream-side wait: the buffer about to be reused has been filled
                pending.append(dist.all_gather_into_tensor(gather_bufs[i % 2], packed, async_op=True))
                last["packed"], last["buf"] = packed, gather_bufs[i % 2]
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

    # ---- N > 1: the gathered buffer of the last step holds every rank's rows, in rank order (outside the timed region) ---
    gather_check = None
    if world > 1:
        mine = last["packed"].double().sum(dim=0).cpu()  # [6] column sums of this rank's per-ray outputs
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        got = last["buf"].view(world, R, 6).double().sum(dim=1).cpu()
        ok = all(torch.allclose(got[r], sums[r], rtol=1e-9, atol=1e-6) for r in range(world))
        distinct = len({tuple(round(float(v), 3) for v in s_) for s_ in sums}) == world  # ranks rendered different batches
        if not ok:
            raise SystemExit(f"rank {rank}: the all-gathered buffer does not hold the ranks' outputs in rank order")
        gather_check = {"ranks_in_buffer": world, "rows_per_rank": R, "bytes_per_rank": R * 6 * 4, "distinct_batches": distinct}

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream -------------------------------
    roofline = None
    extra = {}
    if rank == 0:
        avg = launch_time(lambda i: ops.render_rays(fh, scene_u, opts_for(batches[i % DISTINCT_BATCHES][5]),
                                                    *batches[i % DISTINCT_BATCHES][:4]), min(args.steps, 20))
        alg_bytes = R * (S * BYTES_PER_SAMPLE + BYTES_PER_RAY_IO)
        achieved = alg_bytes / avg / 1e9
        pmc = pmc_summary()
        kernel_name = ("render_fused_kernel<false,false>" if os.environ.get("CN_FUSED_SPLIT", "1") == "0"
                       else "render_split_kernel")
        split = mp == L.MATRIX_SPLIT_BF16
        # matrix work executed: the split mode issues three bf16 products per fp32 product (hi.hi + hi.lo + lo.hi)
        mlp_tflops = R * S * 2 * MLP_MAC_PER_SAMPLE * (3 if split else 1) / avg / 1e12
        mfma_peak = MFMA_F16_PEAK_TFLOPS if split else MFMA_FP32_PEAK_TFLOPS
        limited = ("SIMD issue: VALU cycles (hashing, blending, the operand splits, compositing) + bf16 MFMA cycles, and the L1's "
                   "line-lookup rate of the corner gathers; not HBM" if split else
                   "SIMD issue: fp32 MFMA cycles + VALU cycles (they add on a SIMD); not HBM")
        # `frac` = ALGORITHMIC bytes (1024 B per field sample + 68 B per ray) / launch time / HBM peak, the
        # figure the target is stated in.  The kernel is not limited by HBM: the 64 MB table is cache-resident (measured
        # traffic below is a small fraction of the algorithmic bytes).  What limits it is SIMD issue:
        # `limited_by` says so and `roofline_mfma` prices the matrix half of it.
        # `bound` keeps the key the contract names; the kernel is NOT HBM-bound: `limited_by` / `bound_measured` say what it is
        roofline = {"bound": "hbm", "bound_measured": "simd-issue + L1 line lookups" if split else
                    "simd-issue (fp32 MFMA + VALU cycles add on a SIMD)", "kernel": kernel_name,
                    "matrix_precision": mp_name,
                    "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": pmc.get("traffic"), "traffic_source": pmc.get("source"),
                    "hbm_measured_frac": (round(pmc["traffic"] / avg / 1e9 / HBM_PEAK_GBPS, 4) if pmc.get("traffic") else None),
                    "limited_by": limited,
                    "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_sample": BYTES_PER_SAMPLE,
                    "avg_launch_ms": round(avg * 1e3, 4), "mfma_frac": round(mlp_tflops / mfma_peak, 4)}
        extra["roofline_mfma"] = {"bound": "mfma", "kernel": kernel_name, "achieved": round(mlp_tflops, 2),
                                  "peak": mfma_peak, "unit": "TFLOP/s",
                                  "frac": round(mlp_tflops / mfma_peak, 4), "traffic": None,
                                  "flops_per_sample": 2 * MLP_MAC_PER_SAMPLE * (3 if split else 1),
                                  "note": ("three bf16 products per fp32 product (v_mfma_f32_16x16x32_bf16, operands hi + lo); "
                                           "dense bf16 matrix peak" if split else
                                           "exact-fp32 matrix products (v_mfma_f32_16x16x4_f32); dense fp32 matrix peak")}
        extra["uniform_samples_per_sec_single_launch"] = R * S / avg

    # ---- secondary numbers (rank 0, N=1): proposal-mode render and training iterations --------------------------------
    if rank == 0 and world == 1 and not args.no_secondary:
        extra.update(secondary_timings(args, cfg, params, fspec, batches, ops, fh, dh, scene_c, opts))
        if not args.no_subsystems:
            extra.update(subsystem_timings(args, params, device))

    # ---- CPU baseline (rank 0, N=1): the oracle ("port") on a bounded sample of the same workload ---------------
    # LAST: its 16 compute threads keep spinning for a while after every parallel region, and the per-launch GPU timings above
    # include the host's enqueue gaps (measured: the fp16-mode line read 1.47 ms after the baseline and 1.39-1.42 ms without it)
    cpu_baseline = None
    psnr = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, psnr = run_cpu_baseline(args, params, fspec, pspecs, batches, ops, fh, scene_u, opts)

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
            "dtype": {"fp32": "f32", "f16": "f16 products, f32 accumulate"}.get(
                mp_name, "f32 (matrix products: bf16 hi+lo operand split, f32 accumulate)"),
            "data": "synthetic",
            "config": {"workload": "plant_1-shaped synthetic scene (P-rand), 800x800, 192 samples/ray, 65536-ray batch"
                                   f", mode M-{args.mode}, fused cn_render_rays, eval (no jitter)",
                       "rays_per_batch": R, "samples_per_ray": S, "image": [H, W], "mode": args.mode,
                       "matrix_precision": mp_name + " (FruitNerfModelConfig default; exact fp32: secondary.exact_fp32_ms)",
                       **({"variant": args.variant + " (profiling aid, not the headline)"} if args.variant != "headline" else {}),
                       "sharding": "ray batches per rank + RCCL all-gather of per-ray outputs" if world > 1 else "none"},
            "rays_per_sec": world * R * args.steps / elapsed,
            "psnr_vs_oracle_db": psnr,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        if gather_check is not None:
            line["gather_check"] = gather_check
        emit(line, extra)
    if world > 1:
        dist.destroy_process_group()


MAX_LINE_BYTES = 4096  # the driver reads the LAST stdout line; round 4's 20.6 KB line was not parsed (12.0 KB still was)


def _num(x, digits=4):
    """A bare number for the flat `secondary` object: 4 significant digits are what the timings carry."""
    if x is None or isinstance(x, (bool, str)):
        return x
    return float(f"{float(x):.{digits}g}")


def flat_secondary(extra: dict) -> dict:
    """Bare numbers of the secondary workloads (the full records, with their rooflines, workload strings and spreads, are in
    bench_secondary.json next to this file)."""
    f = {}

    def put(key, *path, scale=1.0):
        d = extra
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return
            d = d[k]
        if isinstance(d, (int, float)):
            f[key] = _num(d * scale)

    put("mfma_tflops", "roofline_mfma", "achieved")
    put("exact_fp32_ms", "uniform_mode_exact_fp32", "ms_per_batch")
    put("exact_fp32_frac", "uniform_mode_exact_fp32", "roofline", "frac")
    put("psnr_vs_exact_fp32_db", "uniform_mode_exact_fp32", "psnr_of_the_headline_vs_this_render_db")
    put("tcnn_f16_table_ms", "uniform_mode_tcnn_f16_table", "ms_per_batch")
    put("f16_mode_ms", "uniform_mode_tcnn_f16_mfma", "ms_per_batch")
    put("f16_mode_samples_per_s", "uniform_mode_tcnn_f16_mfma", "samples_per_sec")
    put("f16_mode_frac", "uniform_mode_tcnn_f16_mfma", "roofline", "frac")
    put("proposal_mode_ms", "proposal_mode", "ms_per_batch")
    put("proposal_sampler_frac_l2", "proposal_mode", "roofline", "frac")
    for key in ("4096", "65536", "65536x192"):
        put(f"train_{key}_ms", "train_iteration", key, "ms_per_iter")
        put(f"train_{key}_fresh_batches_ms", "train_iteration", key, "fresh_batches", "ms_per_iter")
        put(f"train_{key}_frac_atomic", "train_iteration", key, "roofline", "frac")
        put(f"train_{key}_f16_ms", "train_iteration", key, "mixed_precision", "ms_per_iter")
    put("train_4096_steady_ms", "train_iteration", "4096", "steady_state", "ms_per_iter")
    put("c4_seconds", "export_pointcloud_c4", "seconds_to_10M_points")
    put("c4_rays_per_s", "export_pointcloud_c4", "rays_per_sec")
    put("c4_frac", "export_pointcloud_c4", "roofline", "frac")
    put("c4_normals_seconds", "export_pointcloud_c4", "normals", "seconds")
    put("dense_export_samples_per_s", "dense_export", "field_samples_per_sec")
    put("dense_export_frac", "dense_export", "roofline", "frac")
    put("projection_job_ms", "projection_job", "ms_per_job")
    put("projection_jobs_per_s", "projection_run", "batched", "jobs_per_sec")
    put("projection_jobs_per_s_png", "projection_run", "batched", "with_png_tree", "jobs_per_sec")
    put("projection_loop_jobs_per_s", "projection_run", "per_job_loop", "jobs_per_sec")
    put("eval_image_800_ms", "eval_image_800", "ms_per_image")
    put("eval_image_800_f16_ms", "eval_image_800", "tcnn_f16_mode", "ms_per_image")
    put("eval_image_800_frac", "eval_image_800", "roofline", "frac")
    put("eval_image_1920x1440_ms", "eval_image_1920x1440", "ms_per_image")
    put("big_render_65536_ms", "generic_shapes", "big_render_ms")
    put("big_train_8192_ms", "generic_shapes", "big_train_ms")
    return f


def compact_line(line: dict, extra: dict) -> dict:
    """The ONE stdout line: contract keys, `config`, a trimmed `roofline` and `cpu_baseline`, a flat `secondary` of bare numbers."""
    out = dict(line)
    rf = line.get("roofline")
    if rf:
        out["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel",
                                              "avg_launch_ms", "mfma_frac", "hbm_measured_frac") if k in rf}
        out["roofline"]["limited_by"] = ("SIMD issue (VALU + bf16 MFMA) and L1 line lookups, not HBM: the table is cache-resident"
                                         if rf.get("matrix_precision") == "split_bf16" else
                                         "SIMD issue (fp32 MFMA + VALU), not HBM: the table is cache-resident")
        if "matrix_precision" in rf:
            out["roofline"]["matrix_precision"] = rf["matrix_precision"]
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _num(cb["value"], 6), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb["sample_short"], "rays_per_sec": _num(cb["rays_per_sec"]),
                               "c1": {"value": _num(cb["c1"]["value"], 6), "unit": "samples/s", "sample": cb["c1"]["sample_short"]}}
    sec = flat_secondary(extra)
    if sec:
        out["secondary"] = sec
        out["secondary_file"] = "bench_secondary.json"
    return out


def emit(line: dict, extra: dict):
    """Full record -> bench_secondary.json (next to this file); the compact line -> stdout, LAST, under 4 KB."""
    full = dict(line)
    full.update(extra)
    text = json.dumps(full)
    try:
        with open(os.path.join(ROOT, "bench_secondary.json"), "w") as fh_:
            fh_.write(text + "\n")
    except OSError as e:
        print(f"bench.py: could not write bench_secondary.json: {e}", file=sys.stderr)
    # stderr stays short too: the driver's captured tail holds stdout THEN stderr, cut to its last few KB
    print(f"bench.py: full record ({len(text)} B) written to bench_secondary.json", file=sys.stderr, flush=True)
    short = json.dumps(compact_line(line, extra))
    if len(short) >= MAX_LINE_BYTES:  # never let the driver's line grow past what it reads: drop the secondaries first
        c = compact_line(line, extra)
        c.pop("secondary", None)
        short = json.dumps(c)
    assert len(short) < MAX_LINE_BYTES, len(short)
    print(short, flush=True)


def tcnn_f16_field(params, device):
    """The reference's default module implementation on the bench scene: tcnn grid geometry with half2 table entries (what a
    reference-trained checkpoint imports to), a seeded random table of that layout, the same MLPs."""
    from cropnerf_amd import config as _PC
    from cropnerf_amd import ops

    tcfg_t = _PC.FruitNerfModelConfig(num_nerf_samples_per_ray=S, implementation="tcnn")
    tspec = tcfg_t.field_spec(num_images=NUM_CAMERAS)
    gq = torch.Generator(device="cpu").manual_seed(0)
    packed = ((torch.rand(2 * tspec.grid.num_packed_entries, generator=gq) * 2 - 1) * 0.1).to(device)
    pt = dict(params)
    pt["field.mlp_base_grid.hash_table"] = ops.tcnn_grid_pack(tspec.grid, packed, torch.float16)
    return ops.FieldHandle(pt, tspec), tspec


def variant_field(variant, params, fspec, fh, device):
    from cropnerf_amd import _lib as L

    if variant == "split_bf16":  # (= the headline since round 5)
        return fh, {"matrix_precision": L.MATRIX_SPLIT_BF16}
    if variant == "exact_fp32":  # the headline of rounds 1-4: v_mfma_f32_16x16x4_f32 products
        return fh, {"matrix_precision": L.MATRIX_FP32}
    fht, _ = tcnn_f16_field(params, device)
    if variant == "tcnn_f16_mfma":
        return fht, {"matrix_precision": L.MATRIX_F16}
    return fht, {}


def launch_time(fn, n: int) -> float:
    """Average seconds per call of ``fn(i)``: HIP events on the current stream (the stream the kernels are launched on)."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    durs = []
    fn(0)
    torch.cuda.synchronize()
    for i in range(n):
        ev0.record()
        fn(i)
        ev1.record()
        ev1.synchronize()
        durs.append(ev0.elapsed_time(ev1) * 1e-3)
    return sum(durs) / len(durs)


def launch_stats(fn, n: int = 30, warm: int = 5) -> dict:
    """Per-CALL HIP-event times of ``fn(i)`` (seconds): median, mean, min, max, sigma over ``n`` calls after ``warm``
    warm-ups.  Every secondary number of this file is a median of these: round 2 timed each secondary mode with ONE event
    pair around 20 calls after a single warm-up, so a host-side stall inside that 50 ms window was counted as kernel time
    (2.65 ms in five runs, 3.1 in two, 3.79 on the driver's box for the same kernel, whose rocprofv3 trace is 2.61 +- 0.11)."""
    import statistics

    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        evs[i][0].record()
        fn(warm + i)
        evs[i][1].record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e-3 for a, b in evs]
    return {"median": statistics.median(t), "mean": sum(t) / n, "min": min(t), "max": max(t), "sigma": statistics.pstdev(t),
            "launches": n, "warmups": warm}


def _ms(st: dict) -> dict:
    return {"ms_per_batch": round(st["median"] * 1e3, 3),
            "spread_ms": {"min": round(st["min"] * 1e3, 3), "mean": round(st["mean"] * 1e3, 3), "max": round(st["max"] * 1e3, 3),
                          "sigma": round(st["sigma"] * 1e3, 3), "launches": st["launches"], "warmups": st["warmups"]}}


def pmc_variant_summary(tag_part: str) -> dict:
    """Counters of a render VARIANT from the newest profiles/r*<tag_part>*_pmc_variant.json (tools/collect_pmc_variant.sh)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*{tag_part}*_pmc_variant.json")))
    if not files:
        return {}
    try:
        with open(files[-1]) as fh_:
            d = json.load(fh_)
        val = lambda k: d[k]["avg_per_launch"] if k in d else None
        out = {"source": f"profiles/{os.path.basename(files[-1])}" + (f" @ {d['commit']}" if d.get("commit") else "")}
        if val("FETCH_SIZE") is not None and val("WRITE_SIZE") is not None:
            out["traffic"] = int((val("FETCH_SIZE") + val("WRITE_SIZE")) * 1024)
        if val("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
            out["l1_line_lookups"] = int(val("TCP_TOTAL_CACHE_ACCESSES_sum"))
        return out
    except Exception as e:  # noqa: BLE001
        return {"source": f"unreadable: {e}"}


def pmc_summary(pattern: str = "r*_pmc_render*.json") -> dict:
    """Memory-side bytes per launch of the render kernel from the committed rocprofv3 PMC passes of this same command
    (``profiles/r*_pmc_render*.json``, written by tools/collect_pmc.sh + tools/summarise_pmc.py: FETCH_SIZE and WRITE_SIZE,
    KiB, separate passes; the summary records the commit it was measured on).  Counters cannot be read from inside this
    process, so the figure is labelled with its source instead of passing as a live measurement.  No 2x FETCH_SIZE
    correction: the guide calibrates that factor for 16-B streaming reads, these are 8-B gathers, and TCC_MISS x 64 B of
    the same run agrees with the raw value.  Infinity-Cache hits are included (an upper bound on HBM bytes)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return {"traffic": None, "source": "no PMC summary under profiles/"}
    try:
        with open(files[-1]) as fh_:
            d = json.load(fh_)
        val = lambda k: d[k]["avg_per_launch"] if isinstance(d[k], dict) else d[k]
        src = f"profiles/{os.path.basename(files[-1])}"
        if d.get("commit"):
            src += f" @ {d['commit']}"
        return {"traffic": int((val("FETCH_SIZE") + val("WRITE_SIZE")) * 1024), "source": src}
    except Exception as e:  # noqa: BLE001
        return {"traffic": None, "source": f"unreadable PMC summary: {e}"}


def secondary_timings(args, cfg, params, fspec, batches, ops, fh, dh, scene_c, opts):
    """Not the headline: the same M-uniform workload through the optional matrix modes and the tcnn-layout fp16 table (what a
    reference checkpoint imports to), the M-proposal render (256+96 proposal-net evals + 192 field evals per ray) and the
    training iteration of the default method config.  Every figure is the MEDIAN of per-call HIP-event times after five
    warm-ups (launch_stats), with the spread printed beside it."""
    out = {}
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops as _ops

    scene_u = _ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)

    def hint(start):
        return {} if args.no_image_hint else {"image_width": W, "pixel_start": start}

    def render(handle, **kw):
        def fn(i):
            o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
            return ops.render_rays(handle, scene_u, ops.render_opts(S, **kw, **hint(start)), o, d, n, f)
        return fn

    def psnr(a, b):
        mse = float(((a - b) ** 2).mean())
        return round(-10.0 * math.log10(max(mse, 1e-30)), 1)

    flops = R * S * 2 * MLP_MAC_PER_SAMPLE
    # ---- exact fp32 matrix products on the headline table (cn_render_opts.matrix_precision = 0: the headline of rounds 1-4) ----
    st = launch_stats(render(fh, matrix_precision=L.MATRIX_FP32))
    t = st["median"]
    alg32 = R * (S * BYTES_PER_SAMPLE + BYTES_PER_RAY_IO)
    pm32 = pmc_summary("r05_pmc_render_fused.json")
    out["uniform_mode_exact_fp32"] = {
        **_ms(st), "samples_per_sec": R * S / t, "rays_per_sec": R / t,
        "psnr_of_the_headline_vs_this_render_db": psnr(render(fh, matrix_precision=L.MATRIX_SPLIT_BF16)(0)["rgb"],
                                                       render(fh, matrix_precision=L.MATRIX_FP32)(0)["rgb"]),
        "roofline": {"bound": "hbm", "kernel": "render_split_kernel<.,fp32,float,generic>", "achieved": round(alg32 / t / 1e9, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg32 / t / 1e9 / HBM_PEAK_GBPS, 4),
                     "traffic": pm32.get("traffic"), "traffic_source": pm32.get("source"),
                     "bytes_per_sample": BYTES_PER_SAMPLE, "mfma_frac": round(flops / t / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4),
                     "limited_by": "SIMD issue: fp32 MFMA cycles + VALU cycles (they add on a SIMD); not HBM"},
        "note": "v_mfma_f32_16x16x4_f32 products: what small batches and training always use, and config matrix_precision='fp32'"}

    # ---- the reference's default module implementation: tcnn grid geometry, half2 table entries (512 B per sample) ------
    fht, tspec = tcnn_f16_field(params, batches[0][0].device)
    alg = R * (S * BYTES_PER_SAMPLE_F16 + BYTES_PER_RAY_IO)
    st = launch_stats(render(fht))
    t = st["median"]
    exact_t = render(fht)(0)["rgb"]
    out["uniform_mode_tcnn_f16_table"] = {
        **_ms(st), "samples_per_sec": R * S / t, "rays_per_sec": R / t,
        "roofline": {"bound": "hbm", "kernel": "render_split_kernel<.,fp32,half,generic>", "achieved": round(alg / t / 1e9, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / t / 1e9 / HBM_PEAK_GBPS, 4),
                     "traffic": pmc_variant_summary("tcnn_f16").get("traffic"),
                     "traffic_source": pmc_variant_summary("tcnn_f16").get("source"),
                     "bytes_per_sample": BYTES_PER_SAMPLE_F16, "mfma_frac": round(flops / t / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4),
                     "limited_by": "SIMD issue (fp32 MFMA + VALU), as the headline"},
        "note": "tcnn-compatible layout (dense coarse levels, +0.5 offset) with fp16 table entries -- what a reference-trained "
                "checkpoint imports to; arithmetic stays exact fp32 on those values"}
    # ---- ... in the reference's OWN arithmetic class: fp16 operands on v_mfma_f32_16x16x32_f16, fp32 accumulation ---------------
    st = launch_stats(render(fht, matrix_precision=L.MATRIX_F16))
    t = st["median"]
    pv = pmc_variant_summary("f16")
    lookups = pv.get("l1_line_lookups")
    out["uniform_mode_tcnn_f16_mfma"] = {
        **_ms(st), "samples_per_sec": R * S / t, "rays_per_sec": R / t,
        "psnr_vs_fp32_render_db": psnr(render(fht, matrix_precision=L.MATRIX_F16)(0)["rgb"], exact_t),
        "roofline": {"bound": "hbm", "kernel": "render_split_kernel<.,f16,half,generic>", "achieved": round(alg / t / 1e9, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / t / 1e9 / HBM_PEAK_GBPS, 4),
                     "traffic": pv.get("traffic"), "traffic_source": pv.get("source"),
                     "bytes_per_sample": BYTES_PER_SAMPLE_F16,
                     "mfma_frac": round(flops / t / 1e12 / MFMA_F16_PEAK_TFLOPS, 5),
                     "mfma_peak_tflops": MFMA_F16_PEAK_TFLOPS,
                     "l1_line_lookups_per_launch": lookups,
                     "l1_lookup_frac": (round(lookups / t / L1_LOOKUPS_PER_SEC, 4) if lookups else None),
                     "limited_by": "the L1's line-lookup rate for per-lane-addressed loads (one per clock and CU) plus two "
                                   "cycles per L1 miss fill; matrix work is 44 fp16 MFMAs per 32 samples (DESIGN.md 4.12)"},
        "note": "cn_render_opts.matrix_precision = CN_MATRIX_F16: tcnn's FullyFusedMLP arithmetic (fruit_field.py:95,125-167, "
                "fruit_nerf_config.py:35 mixed_precision) -- fp16 weights and layer inputs, fp32 accumulation, packed-fp16 "
                "grid interpolation; parity against oracle tcnn_half_activations in tests/test_gpu_f16.py"}

    def prop(i):
        o, d, n, f, cam, start = batches[i % DISTINCT_BATCHES]
        ps = ops.proposal_sample(dh, scene_c, o, d, n, f, cfg.num_proposal_samples_per_ray, S)
        ops.render_rays(fh, scene_c, ops.render_opts(S, matrix_precision=opts.matrix_precision, **hint(start)), o, d, n, f,
                        bins=ps["euclidean_bins"])  # (the product default's arithmetic, as the headline)

    st = launch_stats(prop, 20)
    t = st["median"]
    out["proposal_mode"] = {**_ms(st), "rays_per_sec": R / t, "field_samples_per_sec": R * S / t,
                            "network_evals_per_sec": R * (S + sum(cfg.num_proposal_samples_per_ray)) / t}
    # the sampler kernel on its own: SURVEY.md 8(d) prices a proposal sample at 320 B of table reads
    n_prop = sum(cfg.num_proposal_samples_per_ray)
    tp = launch_stats(lambda i: ops.proposal_sample(dh, scene_c, *batches[i % DISTINCT_BATCHES][:4],
                                                    cfg.num_proposal_samples_per_ray, S), 20)["median"]
    alg = R * n_prop * BYTES_PER_PROPOSAL_SAMPLE
    # The two proposal tables (2 x 5 levels x 2^17 entries x 8 B = 10 MB) stay in the XCDs' L2s, so the algorithmic bytes are
    # priced against the rate the guide measures for rows gathered from the L2 (MI355X_MICROARCH.md, "Indexed rows": 16.8 TB/s
    # chip-wide, its lower figure), not against HBM, whose 8 TB/s this kernel's algorithmic rate exceeds.
    out["proposal_mode"]["roofline"] = {
        "bound": "l2", "kernel": "proposal_sample_kernel", "achieved": round(alg / tp / 1e9, 1), "peak": L2_GATHER_PEAK_GBPS,
        "unit": "GB/s", "frac": round(alg / tp / 1e9 / L2_GATHER_PEAK_GBPS, 4), "traffic": workload_traffic("proposal", R)[0],
        "traffic_source": workload_traffic("proposal", R)[1], "algorithmic_bytes_per_launch": alg, "bytes_per_sample": BYTES_PER_PROPOSAL_SAMPLE, "avg_launch_ms": round(tp * 1e3, 4),
        "limited_by": "one dependent chain per ray (gathers -> MLP on the matrix cores -> compositing scan -> cdf -> inverse cdf, "
                      "twice) at four waves per SIMD; ablation builds: 0.21 ms without the networks, 0.58 ms with the hash encoding"}

    def timed(fn, n):
        import statistics

        fn(0)
        fn(1)
        torch.cuda.synchronize()
        ts = []
        for i in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(i)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        return statistics.median(ts)

    if not args.no_train:
        from cropnerf_amd import config as PC
        from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
        from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
        from cropnerf_amd.rays import RayBundle, SceneBox

        g = torch.Generator().manual_seed(0)
        train = {}
        train_data: dict = {}
        from cropnerf_amd import synthetic
        from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManager
        from cropnerf_amd.rays import Cameras

        dev = batches[0][0].device
        c2w, intr = synthetic.orbit_cameras(NUM_CAMERAS, height=H, width=W, focal=FOCAL)
        cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], H, W).to(dev)
        # (rays per batch, field samples per ray): the reference's default batch, the same at C2's batch size, and C2's
        # training half at its stated size (BASELINE.json configs[1]: 192 samples/ray, 65k-ray batch)
        for nrays, spp in ((4096, 48), (65536, 48), (65536, 192)):
            tcfg = PC.FruitNerfModelConfig(num_nerf_samples_per_ray=spp)  # (256, 96) proposal samples
            model = FruitModel(tcfg, SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), NUM_CAMERAS,
                               {"semantics": Semantics()}, device=dev, params={k: v.clone() for k, v in params.items()})
            model.training = True
            tr = FruitTrainer(model)  # a fresh trainer: the 7 iterations below are steps 0..6, all of which update the proposal nets
            # training batches are random pixels over all images (PixelSampler), not a block of one image
            idx = torch.stack([torch.randint(0, NUM_CAMERAS, (nrays,), generator=g), torch.randint(0, H, (nrays,), generator=g),
                               torch.randint(0, W, (nrays,), generator=g)], -1)
            idx = idx.to(dev)
            if nrays >= FruitDataManager.SORT_BATCHES_FROM:  # as the data manager hands such a batch out (round 5): by camera and pixel
                idx = idx[ops.ray_sort_permutation(idx, H, W)]
            rb = cams.generate_rays(idx)
            batch = {"image": torch.rand(nrays, 3, generator=g), "fruit_mask": (torch.rand(nrays, 1, generator=g) > 0.5).float()}
            batch = {k: v.to(dev) for k, v in batch.items()}
            t = timed(lambda i: tr.train_iteration(rb, batch), 5)
            key = str(nrays) if spp == 48 else f"{nrays}x{spp}"
            train[key] = {"ms_per_iter": round(t * 1e3, 3), "rays_per_sec": nrays / t, "field_samples_per_ray": spp,
                          "field_samples_per_sec": nrays * spp / t}
            # The iteration's bound is the rate at which the memory side takes float-atomic requests (hash-grid gradient
            # scatter), not HBM bytes or MFMA.  Requests per iteration are MEASURED (TCC_ATOMIC) for each of the
            # three configurations.
            req = ATOMIC_REQUESTS_PER_ITERATION.get(f"{nrays}x{spp}", 0.0)
            src = ATOMIC_REQUESTS_SOURCE
            t_bytes, t_src = workload_traffic(f"train_{nrays}x{spp}", 1)
            train[key]["roofline"] = {
                "bound": "hbm", "kernel": "train_iteration (field_backward_mfma_kernel + 2 x proposal_backward_wave_kernel + 3 x cell_scatter_fold_blocks_kernel)",
                "achieved": round(req / t / 1e9, 3), "peak": round(ATOMIC_REQUESTS_PER_SEC / 1e9, 2),
                "unit": "G atomic requests/s (memory side; 64-byte read-modify-writes)", "frac": round(req / t / ATOMIC_REQUESTS_PER_SEC, 4),
                "traffic": t_bytes, "traffic_source": t_src, "atomic_requests_per_iteration": int(req),
                "requests_source": src,
                "hbm_equivalent_GBps": round(req * 64 / t / 1e9, 1),
                "limited_by": "the field backward (half of the iteration): its float-atomic requests share one port per XCD "
                              "(~1 request per clock for 32 CUs) and its tile is a chain of 17 barrier-separated matrix phases at two "
                              "waves per SIMD -- 74-80 % of its own request floor at 48 samples per ray, the phase chain alone at 192; "
                              "the fraction priced here falls when requests are removed faster than time (DESIGN.md 4.17, 7)"}
            # The figure above hands train_iteration the SAME resident tensors every time: the replayed graph then skips its
            # input copies, and ray generation and the pixel gather sit outside the timed region -- a lower bound on a real
            # iteration.  Here every iteration gets a fresh batch from the data manager (FruitDataManager.next_train: pixel
            # draws on the host, one asynchronous index copy, cn_raygen_pinhole, the image / mask gathers from resident
            # images) on a fresh trainer: iterations 0-1 warm up and capture, 2-9 are timed on the wall clock with no wait
            # between them (all ten update the proposal networks, like the five timed above).
            from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManager, FruitDataManagerConfig

            if "images" not in train_data:
                train_data["images"] = torch.rand(NUM_CAMERAS, H, W, 3, device=dev)
                train_data["masks"] = (torch.rand(NUM_CAMERAS, H, W, 1, device=dev) > 0.5).float()
            dm = FruitDataManager(FruitDataManagerConfig(train_num_rays_per_batch=nrays), cams, device=dev,
                                  images=train_data["images"], fruit_masks=train_data["masks"], seed=3)
            model2 = FruitModel(tcfg, SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), NUM_CAMERAS,
                                {"semantics": Semantics()}, device=dev, params={k: v.clone() for k, v in params.items()})
            model2.training = True
            tr2 = FruitTrainer(model2)
            for i in range(2):
                tr2.train_iteration(*dm.next_train(i))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(2, 10):
                tr2.train_iteration(*dm.next_train(i))
            torch.cuda.synchronize()
            tf = (time.perf_counter() - t0) / 8
            train[key]["fresh_batches"] = {
                "ms_per_iter": round(tf * 1e3, 3), "rays_per_sec": nrays / tf,
                "timing": "wall clock over 8 consecutive iterations, each on a new batch from FruitDataManager.next_train (host pixel "
                          "draws, index copy, ray generation, image / mask gathers, the graph's input copies) -- what a training run "
                          "pays; ms_per_iter above is the same iteration on one resident batch (device events)"}
            del tr2, model2, dm
            # The same iteration in the reference's training arithmetic (mixed_precision=True on tiny-cuda-nn's fp16 modules,
            # fruit_nerf_config.py:35): matrix_precision="f16" -- fp16 operands in the field's forward and backward recompute,
            # bf16 gradient products, fp32 sums and master parameters (cn_field_backward_mp).  Reported beside the exact-fp32
            # figure, which stays the one the rooflines above are priced on.
            tcfg16 = PC.FruitNerfModelConfig(num_nerf_samples_per_ray=spp, matrix_precision="f16")
            model3 = FruitModel(tcfg16, SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), NUM_CAMERAS,
                                {"semantics": Semantics()}, device=dev, params={k: v.clone() for k, v in params.items()})
            model3.training = True
            tr3 = FruitTrainer(model3)
            t16 = timed(lambda i: tr3.train_iteration(rb, batch), 5)
            train[key]["mixed_precision"] = {
                "ms_per_iter": round(t16 * 1e3, 3), "rays_per_sec": nrays / t16,
                "arithmetic": "field forward and backward recompute: fp16 operands (v_mfma_f32_16x16x32_f16 / 16x16x16_f16); "
                              "gradient products: bf16 operands; fp32 accumulation, fp32 master parameters, Adam in fp32; "
                              "proposal networks fp32"}
            del tr3, model3
            if spp == 48:
                # The iterations above are the first seven of a run, where the reference updates the proposal networks every
                # time.  The schedule (fruit_nerf.py:144-149: update_every = 5 once step >= proposal_warmup = 5 000) makes that
                # one iteration in five for most of a run: the same trainer moved to step 5 000, 25 iterations (five schedule
                # periods) on the wall clock with no wait between them.
                tr.step = tr._sampler_step = 5000
                tr._steps_since_update = 0
                for i in range(15):
                    tr.train_iteration(rb, batch)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(25):
                    tr.train_iteration(rb, batch)
                torch.cuda.synchronize()
                ts = (time.perf_counter() - t0) / 25
                train[key]["steady_state"] = {
                    "ms_per_iter": round(ts * 1e3, 3), "rays_per_sec": nrays / ts,
                    "schedule": "step >= proposal_warmup (5 000): the proposal networks take a gradient every 5th iteration",
                    "timing": "wall clock over 25 consecutive iterations"}
            del tr, model
        out["train_iteration"] = train
    return out


def subsystem_timings(args, params, device):
    """The other three subsystems the north star replaces, at the reference's own call shapes (driver-timed: round 2 only had
    builder probes under tools/).  Outside the headline's timed region; each with its algorithmic bytes and fraction of HBM
    peak.  Same synthetic scene (P-rand) with a density / fruit-logit offset so that the exporters keep points."""
    from cropnerf_amd import config as PC
    from cropnerf_amd import ops, synthetic
    from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig
    from cropnerf_amd.fruit_nerf.export.exporter_utils import sample_volume
    from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import generate_point_cloud
    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context
    from cropnerf_amd.fruit_nerf.fruit_pipeline import FruitPipeline, FruitPipelineConfig
    from cropnerf_amd.rays import Cameras, SceneBox

    out = {}
    cfg = PC.FruitNerfModelConfig()  # the default method: (256, 96) proposal samples + 48 field samples per ray
    p2 = {k: v.clone() for k, v in params.items()}
    p2["field.mlp_base_mlp.layers.1.bias"][0] += 4.0  # some density ...
    p2["field.field_head_semantics.net.bias"] += 3.0   # ... and fruit, so the exporters keep points
    c2w, intr = synthetic.orbit_cameras(NUM_CAMERAS, height=H, width=W, focal=FOCAL)
    cams = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], H, W)
    box = SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
    n_prop = sum(cfg.num_proposal_samples_per_ray)
    bytes_per_ray_default = n_prop * BYTES_PER_PROPOSAL_SAMPLE + cfg.num_nerf_samples_per_ray * BYTES_PER_SAMPLE

    def wall(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    # Algorithmic bytes of a default-method ray: the field's 48 x 1 024 B come from the 64 MB table (priced against HBM, like the
    # headline), the 352 x 320 B of the proposal networks from 10 MB that stay in the L2s (priced against the guide's L2 gather
    # rate).  `frac` = the time those two rates allow / the measured time.
    field_bytes_per_ray = cfg.num_nerf_samples_per_ray * BYTES_PER_SAMPLE
    prop_bytes_per_ray = n_prop * BYTES_PER_PROPOSAL_SAMPLE

    def mixed_roofline(rays_, t_, kernel, limited_by, pmc=None):
        """`pmc`: the workload of tools/pmc_workloads.py whose profiled memory-side bytes per ray price `traffic`."""
        bound_t = rays_ * (field_bytes_per_ray / (HBM_PEAK_GBPS * 1e9) + prop_bytes_per_ray / (L2_GATHER_PEAK_GBPS * 1e9))
        traffic, tsrc = workload_traffic(pmc, rays_) if pmc else (None, None)
        r = {"bound": "hbm+l2", "kernel": kernel, "achieved": round(rays_ * bytes_per_ray_default / t_ / 1e9, 1), "unit": "GB/s",
             "peak": {"field_bytes_hbm": HBM_PEAK_GBPS, "proposal_bytes_l2": L2_GATHER_PEAK_GBPS},
             "frac": round(bound_t / t_, 4), "traffic": traffic, "bytes_per_ray": bytes_per_ray_default,
             "field_bytes_per_ray": field_bytes_per_ray, "proposal_bytes_per_ray": prop_bytes_per_ray, "limited_by": limited_by}
        if traffic is not None:
            r["traffic_source"] = tsrc
            r["memory_side_GBps"] = round(traffic / t_ / 1e9, 1)
        return r

    # ---- ns-export pointcloud (BASELINE.json configs[3]): random training rays until 10 M points are kept ---------------------
    # exporter_utils_nerfacto.py:125-183.  Call sizes: 2 048 rays (the reference's copy, debug/exporter_nerfacto.py:91) and
    # 32 768 (upstream ns-export, README.md:125); no outlier removal in the timed part.  The scene keeps a few per cent of the
    # rays (a boll fills little of a frame): the semantic head's bias is set, on a sample of the exporter's own rays, so that
    # ~3 % of them end above the 0.9 threshold -- the compaction rejects 97 %, 10 M points take ~3.3e8 rays.
    def keep_fraction_bias(target_frac):
        pipe_c = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(2048, 2048), cfg), device, cams, box, test_mode="test",
                               params={k: v.clone() for k, v in p2.items()})
        first = torch.zeros(1, dtype=torch.int64, device=device)
        idx = ops_.pixel_sample(pipe_c.datamanager.export_seed, first, 32, 2048, NUM_CAMERAS, H, W)
        o_ = pipe_c.model(pipe_c.datamanager.cameras.generate_rays(idx))
        sem, acc = o_["semantics"][:, 0].double(), o_["accumulation"][:, 0].double()
        lo_b, hi_b = -20.0, 20.0  # composited logit = sum w * (logit + b) = sem + b * acc: monotone in b
        for _ in range(40):
            mid = 0.5 * (lo_b + hi_b)
            if float(((sem + mid * acc) > math.log(9.0)).double().mean()) > target_frac:
                hi_b = mid
            else:
                lo_b = mid
        return 0.5 * (lo_b + hi_b)

    from cropnerf_amd import ops as ops_

    bias = keep_fraction_bias(0.03)
    p_c4 = {k: v.clone() for k, v in p2.items()}
    p_c4["field.field_head_semantics.net.bias"] += bias

    def export_case(rays_per_call, launch_rays, num_points):
        pipe_x = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(rays_per_call, rays_per_call), cfg), device, cams, box,
                               test_mode="test", params=p_c4)
        generate_point_cloud(pipe_x, num_points=20_000, remove_outliers=False, launch_rays=launch_rays)  # warm-up / graph capture
        st = {}
        t_, pcd_ = wall(lambda: generate_point_cloud(pipe_x, num_points=num_points, remove_outliers=False, launch_rays=launch_rays,
                                                     stats=st))
        kept_ = int(pcd_["points"].shape[0])
        last_cloud["points"] = pcd_["points"]
        return {"seconds": round(t_, 3), "kept_points": kept_, "calls": st["calls"], "rays_per_call": rays_per_call,
                "calls_per_launch": st["calls_per_launch"], "rays_rendered": st["rays"], "kept_fraction": round(kept_ / st["rays"], 4),
                "rays_per_sec": st["rays"] / t_, "points_per_sec": kept_ / t_, "seconds_to_10M_points": round(t_ * 1e7 / kept_, 2),
                "roofline": mixed_roofline(st["rays"], t_, "pixel_sample + raygen + proposal_sample_kernel + render kernel + "
                                           "pointcloud_compact_calls per launch",
                                           f"{st['calls_per_launch']} call(s) of {rays_per_call} rays per launch sequence"
                                           + (" (HIP-graph replay)" if st.get("graph") else "") + "; RANDOM pixels of random cameras, "
                                           "rendered sorted by camera and pixel inside a launch: what is left of the field's L2 misses "
                                           "(draw order: ~108 KB of memory-side traffic per ray against 3 KB for an image's coherent rays) "
                                           "bounds it", pmc="export_c4")}

    last_cloud: dict = {}
    c4 = export_case(2048, 1 << 20, 10_000_000)
    # the two post-processing passes of `ns-export pointcloud` on that cloud (exporter_utils_nerfacto.py:194-225): statistical
    # outlier removal (20 neighbours) and open3d-style normals (30 neighbours, covariance, smallest eigenvector) + re-orientation,
    # both on the device (cn_knn_mean_distance / cn_estimate_normals)
    pts_c4 = torch.from_numpy(last_cloud["points"]).to(device=device, dtype=torch.float32)
    ops.estimate_normals(pts_c4[:100000], 30)
    t_out, _ = wall(lambda: ops.statistical_outlier_mask(pts_c4, 20, 10.0))
    t_nrm, nd = wall(lambda: ops.estimate_normals(pts_c4, 30))
    c4["normals"] = {"points": int(pts_c4.shape[0]), "seconds": round(t_nrm, 3), "points_per_sec": pts_c4.shape[0] / t_nrm,
                     "degenerate": int(nd[1].sum()), "outlier_pass_seconds": round(t_out, 3),
                     "kernel": "knn_mean_distance_grid_kernel / knn_normals_grid_kernel on the two-level grid (binning included: torch sort, "
                               "unique, offsets), 30-nearest search + fp64 covariance + closed-form 3x3 eigen-solve"}
    del pts_c4, nd
    c4["call_size_32768"] = export_case(32768, 1 << 20, 10_000_000)
    c4["one_call_per_launch_2048"] = export_case(2048, None, 1_000_000)  # the reference's loop shape, graph-replayed (round 3)
    c4["semantic_bias_for_3pct"] = round(bias, 4)
    c4["workload"] = ("ns-export pointcloud --num-points 10000000 on the synthetic scene with ~3 % of the rays kept, default method (48 "
                      "field + 352 proposal samples per ray); 2 048-ray calls (debug/exporter_nerfacto.py:91) grouped 512 per launch and rendered "
                      "in (camera, pixel Morton) order, the outputs put back in draw order before the compaction (round 5; 32 per "
                      "launch in draw order until then); 32 768-ray calls (upstream ns-export) 32 per launch; the one-call-per-launch "
                      "loop on 1 M points beside them; "
                      "reference: exporter_utils_nerfacto.py:125-183")
    out["export_pointcloud_c4"] = c4
    pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(2048, 2048), cfg), device, cams, box, test_mode="test", params=p2)
    # ---- exporter.py semantic-pointcloud (dense volume export): 512-ray calls x 3 000 samples per ray -----------------------------
    # scripts/exporter.py:75-77, exporter_utils.py:93-172; the reference's full job is 3000 x 3000 rays, timed here on 512 x 512
    aabb_e = ((-1, -1, -1 + .318), (1, 1, 1 + .318))
    side = 512

    def dense(params_e):
        pipe_e = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(4096, 512), cfg), device, cams, box, test_mode="export",
                               params=params_e)
        pipe_e.model.setup_inference(True, 3000)
        n_small = pipe_e.datamanager.setup_inference(aabb_e, 64)
        sample_volume(pipe_e, n_small, transform_json={"scale": 1.0, "transform": None}, capacity=1 << 26)  # warm-up
        n_rays = pipe_e.datamanager.setup_inference(aabb_e, side)
        t, pcds = wall(lambda: sample_volume(pipe_e, n_rays, transform_json={"scale": 1.0, "transform": None}, capacity=1 << 26))
        return t, n_rays, {k: int(v["points"].shape[0]) for k, v in pcds.items()}

    # (a) the scene as it is: no sample passes the exporter's density >= 70 threshold -- the device side alone (render + masks +
    # compaction); (b) a density offset chosen so that ~1 % of the samples pass (the P-rand density logits are -0.095 +- 0.01):
    # the kept points of the three sets then also cross PCIe and become float64 arrays on the host, as the exporter hands them on
    t, n_rays, kept0 = dense(p2)
    p3 = {k: v.clone() for k, v in params.items()}
    p3["field.mlp_base_mlp.layers.1.bias"][0] += 4.323
    p3["field.field_head_semantics.net.bias"] += 3.0
    t1, _, kept1 = dense(p3)
    ns = n_rays * 3000
    out["dense_export"] = {
        "seconds": round(t, 3), "rays": n_rays, "samples_per_ray": 3000, "rays_per_call": 512, "field_samples_per_sec": ns / t,
        "kept": kept0, "full_3000x3000_estimate_s": round(t * (3000 * 3000) / n_rays, 2),
        "with_1pct_kept": {"seconds": round(t1, 3), "kept": kept1, "field_samples_per_sec": ns / t1,
                           "note": "the kept points of the three sets (3 x 7 floats each) copied to the host and converted to "
                                   "float64 arrays inside the timed region, as sample_volume returns them"},
        "roofline": {"bound": "hbm", "kernel": "render_split_kernel<per-sample> + export_compact", "achieved": round(ns * BYTES_PER_SAMPLE / t / 1e9, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ns * BYTES_PER_SAMPLE / t / 1e9 / HBM_PEAK_GBPS, 4),
                     "traffic": workload_traffic("dense_export", ns)[0], "traffic_source": workload_traffic("dense_export", ns)[1],
                     "bytes_per_sample": BYTES_PER_SAMPLE,
                     "limited_by": "SIMD issue of the render kernel (as the headline) + 28 B per sample of per-sample outputs"},
        "workload": f"exporter.py semantic-pointcloud: {side} x {side} surface rays (reference: 3000 x 3000) x 3000 samples, 512 rays per "
                    "call; reference: scripts/exporter.py:75-77, export/exporter_utils.py:93-172"}
    # ---- semantic projection: one (camera, sub-cluster AABB) job at 800 x 800, both passes (fruit_nerf.py:283-315) -----------------
    m = pipe.model
    aabb = SceneBox(torch.tensor([[-0.15, -0.15, -0.15], [0.15, 0.15, 0.15]]))
    import statistics

    with background_color_override_context(torch.zeros(3)):
        m.project_cluster(cams[0], aabb, 0)
        ts = []
        for i in range(10):
            tj, _ = wall(lambda: m.project_cluster(cams[i], aabb, i))
            ts.append(tj)
        rays_in = [int((cams[i].to(device).generate_rays(camera_indices=0, keep_shape=True, aabb_box=aabb).nears < 1e10).sum())
                   for i in range(10)]
    tj = statistics.median(ts)
    vr = statistics.median(rays_in)
    out["projection_job"] = {
        "ms_per_job": round(tj * 1e3, 3), "spread_ms": {"min": round(min(ts) * 1e3, 3), "max": round(max(ts) * 1e3, 3), "jobs": 10},
        "image": [H, W], "rays_inside_aabb": int(vr), "passes": 2, "rays_per_sec": 2 * vr / tj,
        "roofline": mixed_roofline(2 * vr, tj, "proposal_sample_kernel + render kernels (full pass, density-only occlusion pass)",
                                   "ray generation + AABB test for all 640 000 pixels, one index list of the rays inside the box, "
                                   "then the two renders of those rays (sampler and 48-sample field pass each)", pmc="projection"),
        "workload": "get_outputs_for_projections, one camera x one sub-cluster AABB (0.3-wide box at the origin), 800 x 800: the "
                    "AABB-restricted render and the occlusion pass; reference: fruit_nerf.py:283-315"}
    # ---- the projection stage as the reference RUNS it (fruit_nerf.py:254-318, scripts/semantic_projection.py:132-170): every
    # (super-cluster, camera, sub-cluster) job -- 8 super-clusters x 16 cameras x 2 boll-halves at 800 x 800 -- with and without
    # the PNG tree, batched (projection.project_all) against the per-job loop that mirrors the reference call for call
    import shutil
    import tempfile

    import numpy as np

    from cropnerf_amd.fruit_nerf.fruit_nerf import Semantics as _Sem

    def boll_boxes(width_lo, width_hi, seed, count):
        """`count` bolls split in two along a random axis, as k-means with k = 2 splits a super-cluster (segmenter.py:153-181)."""
        rng = np.random.default_rng(seed)
        boxes = []
        for _ in range(count):
            c = rng.uniform(-0.3, 0.3, 3)
            half = rng.uniform(width_lo / 2, width_hi / 2, 3)
            ax = int(rng.integers(0, 3))
            lo, hi = c - half, c + half
            mid_lo, mid_hi = hi.copy(), lo.copy()
            mid_lo[ax] = mid_hi[ax] = c[ax]
            boxes.append({"aabb": np.stack([np.stack([lo, mid_lo]), np.stack([mid_hi, hi])]).astype(np.float32)})
        return boxes

    sel = torch.arange(0, NUM_CAMERAS, NUM_CAMERAS // 16)[:16]
    cams16 = Cameras(c2w[sel], intr[sel, 0], intr[sel, 1], intr[sel, 2], intr[sel, 3], H, W)

    class _DS:
        cameras = cams16
        metadata = {"semantics": _Sem()}

    n_sub = 2 * 16 * 2  # the per-job loop is timed on 2 super-clusters

    def projection_case(sc_boxes, what):
        n_jobs = len(sc_boxes) * 16 * 2
        tmp = tempfile.mkdtemp(prefix="cn_proj_")
        try:
            with background_color_override_context(torch.zeros(3)):
                m.get_outputs_for_projections(_DS, None, pcd_data=sc_boxes, save=False, return_run=True)  # warm-up
                t_b, run = wall(lambda: m.get_outputs_for_projections(_DS, None, pcd_data=sc_boxes, save=False, return_run=True))
                m.get_outputs_for_projections(_DS, None, pcd_data=sc_boxes[:1], output_root=os.path.join(tmp, "w"), save=True)
                t_bf, _ = wall(lambda: m.get_outputs_for_projections(_DS, None, pcd_data=sc_boxes,
                                                                   output_root=os.path.join(tmp, "b"), save=True))
                sub = sc_boxes[:2]
                m.get_outputs_for_projections(_DS, None, pcd_data=sub[:1], save=False, batched=False)
                t_p, _ = wall(lambda: m.get_outputs_for_projections(_DS, None, pcd_data=sub, save=False, batched=False))
                t_pf, _ = wall(lambda: m.get_outputs_for_projections(_DS, None, pcd_data=sub, output_root=os.path.join(tmp, "p"),
                                                                   save=True, batched=False))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        rays_run = int(run.stats["rays"])
        return {
            "boxes": what, "jobs": n_jobs, "super_clusters": len(sc_boxes), "rays_inside_boxes": rays_run, "rays_per_job": round(rays_run / n_jobs, 1),
            "rectangle_pixels": int(run.stats["slots"]), "batches": len(run.batches),
            "batched": {"seconds": round(t_b, 4), "jobs_per_sec": round(n_jobs / t_b, 1), "ms_per_job": round(t_b / n_jobs * 1e3, 4),
                        "with_png_tree": {"seconds": round(t_bf, 4), "jobs_per_sec": round(n_jobs / t_bf, 1),
                                          "png_files": 2 * n_jobs, "png_hidden": round(1.0 - (t_bf - t_b) / max(t_bf, 1e-9), 3)}},
            "per_job_loop": {"jobs": n_sub, "seconds": round(t_p, 4), "jobs_per_sec": round(n_sub / t_p, 1),
                             "ms_per_job": round(t_p / n_sub * 1e3, 4),
                             "with_png_tree": {"seconds": round(t_pf, 4), "jobs_per_sec": round(n_sub / t_pf, 1)}},
            "speedup": {"no_files": round((n_jobs / t_b) / (n_sub / t_p), 2),
                        "with_png_tree": round((n_jobs / t_bf) / (n_sub / t_pf), 2)},
            "roofline": mixed_roofline(2 * rays_run, t_b, "projection_test/gather/scatter + proposal_sample_kernel + render kernels",
                                       "two passes over the rays inside the boxes (sampler + 48-sample field pass; sampler + "
                                       "density-only pass), one host synchronisation per batch", pmc="projection")}

    # boll-sized boxes: a 3DCotton boll is 3-5 cm of a plant that fills the +-1 scene box (0.03-0.05 units: ~55 pixels across
    # at 800 x 800 from the 0.8 orbit); the analytic plant of tools/pipeline.py has 0.13-0.17-wide bolls (~200 pixels across)
    small = projection_case(boll_boxes(0.03, 0.05, 5, 32), "0.03-0.05 wide (3DCotton boll scale)")
    large = projection_case(boll_boxes(0.13, 0.17, 6, 8), "0.13-0.17 wide (the analytic plant's bolls)")
    # fixed cost per job: boxes nobody sees (no ray hits: planning, tests and bookkeeping only)
    far = [{"aabb": np.tile(np.array([[[5.0, 5, 5], [5.1, 5.1, 5.1]]], np.float32), (2, 1, 1))} for _ in range(8)]
    with background_color_override_context(torch.zeros(3)):
        t_fb, _ = wall(lambda: m.get_outputs_for_projections(_DS, None, pcd_data=far, save=False, return_run=True))
        t_fp, _ = wall(lambda: m.get_outputs_for_projections(_DS, None, pcd_data=far[:2], save=False, batched=False))
    out["projection_run"] = {
        "cameras": 16, "sub_clusters": 2, "image": [H, W],
        **small, "analytic_plant_boxes": large,
        "fixed_cost_ms_per_job": {"batched": round(t_fb / (8 * 16 * 2) * 1e3, 4), "per_job_loop": round(t_fp / n_sub * 1e3, 4),
                                  "note": "boxes outside every frame: no ray is rendered"},
        "workload": "get_outputs_for_projections: 32 (boll scale) / 8 (analytic-plant scale) super-clusters x 16 cameras x 2 "
                    "sub-cluster boxes at 800 x 800, both passes per job, batched (projection.project_all) against the per-job loop "
                    "that mirrors the reference call for call (timed on 2 of the super-clusters); with_png_tree also writes the reference's file tree (2 PNGs per job); reference: "
                    "fruit_nerf.py:254-318, scripts/semantic_projection.py:132-170"}
    # ---- one whole 800 x 800 eval image of the default method (fruit_nerf.py:377-404; chunks of max(eval_num_rays_per_chunk, 2^18)) ---
    # exact fp32 on the torch-layout model above, and a model as an imported reference checkpoint is -- tcnn layout, fp16 tables --
    # in tcnn's own arithmetic class (matrix_precision = "f16")
    from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics

    def image_ms(model):
        model.get_outputs_for_camera_ray_bundle(cams.to(device).generate_rays(0, keep_shape=True))
        ts_ = []
        for i in range(1, 8):
            ti, _ = wall(lambda: model.get_outputs_for_camera_ray_bundle(cams.to(device).generate_rays(i, keep_shape=True)))
            ts_.append(ti)
        ts_.sort()
        return ts_[len(ts_) // 2], ts_
    ti, ts_ = image_ms(m)
    out["eval_image_800"] = {
        "ms_per_image": round(ti * 1e3, 3), "spread_ms": {"min": round(ts_[0] * 1e3, 3), "max": round(ts_[-1] * 1e3, 3), "images": len(ts_)},
        "rays_per_sec": H * W / ti, "chunk_rays": max(int(cfg.eval_num_rays_per_chunk), int(m.EVAL_CHUNK)),
        "roofline": mixed_roofline(H * W, ti, "proposal_sample_kernel + render_split_kernel per 262 144-ray chunk",
                                   "the sampler (0.6 ms per 65 536 rays) and the 48-sample field pass (two rays per three half-steps since "
                                   "round 4: no empty column tiles), both at SIMD issue", pmc="eval_image"),
        "workload": "get_outputs_for_camera_ray_bundle, default method ((256, 96) proposal + 48 field samples), ray generation "
                    "included; reference: fruit_nerf.py:377-404"}
    # ---- the same at the resolution and from the poses of the reference's one real capture (fruit_nerf/utils/transforms.json:
    # 147 frames, 1920 x 1440, f = 1442.48; camera data = tests/golden/capture_3dcotton.npz, parsed by the dataparser mirror) ---
    cap_path = os.path.join(ROOT, "tests", "golden", "capture_3dcotton.npz")
    if os.path.exists(cap_path):
        from cropnerf_amd.fruit_nerf.data.cotton_nerf_dataparser import CottonNerfDataParserConfig

        fxt = np.load(cap_path)
        cap_dir = tempfile.mkdtemp(prefix="cn_capture_")
        try:
            synthetic.write_transforms_json(cap_dir, fxt["frame_number"], fxt["transform_matrix"], fxt["intrinsics"], fxt["size_hw"],
                                            orientation_override="none", auto_scale_poses_override=False)
            dpo = CottonNerfDataParserConfig(data=cap_dir, downscale_factor=1).setup().get_dataparser_outputs("train")
        finally:
            shutil.rmtree(cap_dir, ignore_errors=True)
        cap_cams = dpo.cameras.to(device)
        Hc, Wc = cap_cams.height, cap_cams.width
        # the capture's 140 training cameras index an appearance / pose table of their own size
        m_cap = FruitModel(cfg, dpo.scene_box, len(cap_cams), {"semantics": Semantics()}, device=device, test_mode="test")

        def cap_image_ms(model):
            model.get_outputs_for_camera_ray_bundle(cap_cams.generate_rays(0, keep_shape=True))
            ts_ = []
            for i in range(1, 6):
                ti_, _ = wall(lambda: model.get_outputs_for_camera_ray_bundle(cap_cams.generate_rays(20 * i, keep_shape=True)))
                ts_.append(ti_)
            ts_.sort()
            return ts_[len(ts_) // 2], ts_

        tc, tsc = cap_image_ms(m_cap)
        out["eval_image_1920x1440"] = {
            "ms_per_image": round(tc * 1e3, 3), "spread_ms": {"min": round(tsc[0] * 1e3, 3), "max": round(tsc[-1] * 1e3, 3), "images": len(tsc)},
            "rays_per_sec": Hc * Wc / tc, "image": [Hc, Wc], "cameras": len(cap_cams),
            "roofline": mixed_roofline(Hc * Wc, tc, "proposal_sample_kernel + render_split_kernel per 262 144-ray chunk",
                                       "as eval_image_800 (the sampler + a 48-sample field pass), 4.3 x the rays", pmc="eval_image"),
            "workload": "get_outputs_for_camera_ray_bundle at the resolution and from the (centred, unit-box-scaled) poses of the "
                        "reference's real capture file, default method, random-init model; reference: fruit_nerf.py:377-404, "
                        "fruit_nerf/utils/transforms.json"}
    cfg16 = PC.FruitNerfModelConfig(implementation="tcnn", hash_table_dtype="float16", matrix_precision="f16")
    m16 = FruitModel(cfg16, box, NUM_CAMERAS, {"semantics": Semantics()}, device=device, test_mode="test")
    t16, ts16 = image_ms(m16)
    out["eval_image_800"]["tcnn_f16_mode"] = {
        "ms_per_image": round(t16 * 1e3, 3), "spread_ms": {"min": round(ts16[0] * 1e3, 3), "max": round(ts16[-1] * 1e3, 3), "images": len(ts16)},
        "note": "tcnn layout, fp16 tables (random values), matrix_precision = f16: what an imported reference checkpoint renders as"}
    del m16
    # ---- the other two method specifications of the plugin (fruit_nerf_config.py:66-172) on the shape-generic kernels ---------------
    # fruit_nerf_method_big: geo 30, 3 x 128 semantic layers, 2^21-entry levels, (512, 256) proposal + 128 field samples per ray,
    # 8 192-ray training batches; fruit_nerf_method_huge: the 7-level second proposal network, 16 384-ray batches, no pose group.
    from cropnerf_amd.fruit_nerf import fruit_nerf_config as FC
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer, groups_from_spec

    gs = {}
    g2 = torch.Generator().manual_seed(5)
    cams8 = Cameras(c2w[:8], intr[:8, 0], intr[:8, 1], intr[:8, 2], intr[:8, 3], H, W).to(device)
    for short, spec_name in (("big", "fruit_nerf_big"), ("huge", "fruit_nerf_huge")):
        tc_ = FC.NATIVE_METHODS[spec_name].config
        mcfg = tc_.pipeline.model
        mg = FruitModel(mcfg, box, 8, {"semantics": Semantics()}, device=device, test_mode="inference")
        for k_, v_ in mg.params.items():
            if k_.endswith("hash_table"):
                v_.mul_(100.0)  # (the 1e-3 initialisation is an empty volume)
        rb_ = cams8.generate_rays(0, keep_shape=False, aabb_box=box)[:65536]
        mg(rb_)
        tr_, _ = wall(lambda: [mg(rb_) for _ in range(3)])
        mg.training = True
        trn = FruitTrainer(mg, groups_from_spec(tc_.optimizers))
        Rt = int(tc_.pipeline.datamanager.train_num_rays_per_batch)
        idx_ = torch.floor(torch.rand(Rt, 3, generator=g2) * torch.tensor([8.0, H, W])).long().to(device)
        rays_t = cams8.generate_rays(idx_)
        batch_t = {"image": torch.rand(Rt, 3, generator=g2).to(device), "fruit_mask": (torch.rand(Rt, 1, generator=g2) > 0.5).float().to(device)}
        for _ in range(2):
            trn.train_iteration(rays_t, batch_t)
        tt_, _ = wall(lambda: [trn.train_iteration(rays_t, batch_t) for _ in range(5)])
        gs[f"{short}_render_ms"] = round(tr_ / 3 * 1e3, 3)
        gs[f"{short}_train_ms"] = round(tt_ / 5 * 1e3, 3)
        gs[f"{short}_train_rays"] = Rt
        gs[f"{short}_samples_per_ray"] = {"proposal": list(mcfg.num_proposal_samples_per_ray), "field": mcfg.num_nerf_samples_per_ray}
        del trn, mg
    gs["workload"] = ("one 65 536-ray inference call and one training iteration at the method's own batch size, shape-generic "
                      "kernels (cn_field_eval / cn_field_backward_general), random-init model with the tables scaled x100")
    out["generic_shapes"] = gs
    return out


def run_cpu_baseline(args, params, fspec, pspecs, batches, ops, fh, scene_u, opts):
    """the CPU oracle (kind "port": this repo's op-for-op PyTorch restatement of the reference's
    path -- the reference itself needs nerfstudio and cannot run) on a bounded sample of the same workload, all host cores
    of this GPU's share: 3 warm-up chunks, then a FIXED number of timed chunks, MEDIAN chunk time.  C2 (800x800 camera,
    192 samples/ray) in 4096-ray chunks -- the number `value` compares with -- and C1 (400x400, 64 samples/ray, 1024-ray
    chunks: BASELINE.json configs[0]).  Also PSNR of the GPU render against the oracle on the C2 sample."""
    import statistics

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
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])

    def timed_chunks(model, rays_cpu, chunk, n_timed):
        o, d, n, f = rays_cpu
        total = o.shape[0]

        def bundle(k):
            sel = (torch.arange(chunk) * (total // chunk) + k) % total  # spread over the batch, shifted per chunk
            return ORY.RayBundle(o[sel], d[sel], torch.zeros(chunk, 1), None, n[sel], f[sel]), sel

        times = []
        with torch.no_grad():
            for k in range(3):
                model.forward(bundle(k)[0])
            for k in range(n_timed):
                rb, _ = bundle(3 + k)
                t0 = time.perf_counter()
                model.forward(rb)
                times.append(time.perf_counter() - t0)
            rb0, sel0 = bundle(0)
            ref = model.forward(rb0)
        return statistics.median(times), sum(times), ref, sel0

    # ---- C2: the bench workload -------------------------------------------------------------------------------------------
    model = OM.OracleModel(cpu_params, OM.ModelConfig(field=ofs, proposals=ops_, disable_scene_contraction=True), aabb,
                           test_mode="inference")
    model.uniform_samples = S
    rays2 = tuple(t.detach().cpu() for t in batches[0][:4])
    chunk2, n2 = 4096, max(10, args.cpu_baseline_chunks)
    med2, tot2, ref, sel = timed_chunks(model, rays2, chunk2, n2)
    gpu = ops.render_rays(fh, scene_u, opts, *(t[sel.to(t.device)].contiguous() for t in batches[0][:4]))
    mse = torch.mean((gpu["rgb"].cpu() - ref["rgb"]) ** 2).item()
    psnr = 10.0 * math.log10(1.0 / max(mse, 1e-20))
    # ---- C1: 400 x 400, 64 samples per ray, 1024-ray chunks -------------------------------------------------------------------
    from cropnerf_amd import synthetic

    dev = batches[0][0].device
    c2w1, intr1 = synthetic.orbit_cameras(1, height=400, width=400, focal=555.6)
    r1 = ops.raygen_pinhole(c2w1.to(dev), intr1.to(dev), cam=0, height=400, width=400, pixel_start=0, num_rays=400 * 400)
    n1_, f1_ = ops.intersect_aabb(r1["origins"], r1["directions"], [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0])
    rays1 = tuple(t.detach().cpu() for t in (r1["origins"], r1["directions"], n1_, f1_))
    model1 = OM.OracleModel(cpu_params, OM.ModelConfig(field=ofs, proposals=ops_, disable_scene_contraction=True), aabb,
                            test_mode="inference")
    model1.uniform_samples = 64
    chunk1, n1 = 1024, 20
    med1, tot1, _, _ = timed_chunks(model1, rays1, chunk1, n1)
    base = {"value": chunk2 * S / med2, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample_short": f"C2: {n2} chunks x {chunk2} rays x {S} spp, median {med2 * 1e3:.0f} ms ({tot2:.1f} s timed), {torch.get_num_threads()} threads",
            "sample": f"C2: {n2} timed chunks of {chunk2} rays x {S} samples of batch 0 after 3 warm-ups, median chunk time "
                      f"{med2 * 1e3:.1f} ms ({tot2:.1f} s timed), torch {torch.get_num_threads()} threads, fp32, eval mode",
            "rays_per_sec": chunk2 / med2,
            "c1": {"value": chunk1 * 64 / med1, "unit": "samples/s", "rays_per_sec": chunk1 / med1,
                   "sample_short": f"C1: 400x400, 64 spp, {n1} chunks x {chunk1} rays, median {med1 * 1e3:.0f} ms",
                   "sample": f"C1 (BASELINE.json configs[0]): 400x400 camera, 64 samples/ray, {n1} timed chunks of {chunk1} rays "
                             f"after 3 warm-ups, median chunk time {med1 * 1e3:.1f} ms ({tot1:.1f} s timed)"}}
    return base, round(psnr, 2)


if __name__ == "__main__":
    main()

"""Where do the occasional slow `secondary` numbers of bench.py come from (round 2: tcnn fp16 table 2.65 ms in five runs,
3.1 in two, 3.79 on the driver's box; split-bf16 2.40 once against 1.89)?

Round 2's bench.py timed each secondary mode ONCE: one warm-up call, then 20 calls between a single pair of HIP events -- so
a host-side stall inside that 52 ms window (Python's cyclic garbage collector, a descheduled launch thread, the caching
allocator going to the driver) leaves the GPU idle and is counted as kernel time.  This probe repeats that window many times,
after the same kind of CPU work bench.py does in front of it, and logs for every window: the single-pair time per call, the
median of per-call event pairs over the same calls, the longest host gap between two launches, and the garbage collections
that ran inside it.

    python tools/secondary_timing_probe.py [--windows 40]
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=40)
    ap.add_argument("--cpu-work", type=float, default=2.0, help="seconds of 16-thread CPU oracle-like work before the windows")
    args = ap.parse_args()
    from cropnerf_amd import ops

    dev = torch.device("cuda", 0)
    cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
    batches = bench.make_batches(ops, c2w, intr, 0, 1)
    scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)
    fht, tspec = bench.tcnn_f16_field(params, dev)
    gc_log = []

    def on_gc(phase, info):
        gc_log.append((time.perf_counter(), phase, info.get("generation"), info.get("collected")))

    gc.callbacks.append(on_gc)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))

    def call(i):
        o, d, n, f, cam, start = batches[i % len(batches)]
        return ops.render_rays(fht, scene, ops.render_opts(bench.S, image_width=bench.W, pixel_start=start), o, d, n, f)

    # CPU work of the kind bench.py's CPU baseline does (many-thread matmuls / gathers), leaving worker threads spinning
    t_end = time.perf_counter() + args.cpu_work
    a = torch.rand(4096 * 192, 64)
    w = torch.rand(64, 64)
    while time.perf_counter() < t_end:
        a = torch.relu(a @ w) * 0.01
    rows = []
    for wdx in range(args.windows):
        if wdx % 4 == 0:  # what bench.py did right before the tcnn window: a big CPU tensor + H2D copy
            g = torch.Generator(device="cpu").manual_seed(wdx)
            junk = ((torch.rand(12_000_000, generator=g) * 2 - 1) * 0.1).to(dev)
            del junk
        call(0)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        gc_before = len(gc_log)
        host = []
        t_prev = time.perf_counter()
        ev0.record()
        for i in range(20):
            pairs[i][0].record()
            call(i)
            pairs[i][1].record()
            t_now = time.perf_counter()
            host.append(t_now - t_prev)
            t_prev = t_now
        ev1.record()
        ev1.synchronize()
        single = ev0.elapsed_time(ev1) / 20
        per = [p[0].elapsed_time(p[1]) for p in pairs]
        gcs = [(g_[2]) for g_ in gc_log[gc_before:] if g_[1] == "start"]
        rows.append({"window": wdx, "single_pair_ms_per_call": round(single, 3), "per_call_median_ms": round(statistics.median(per), 3),
                     "per_call_max_ms": round(max(per), 3), "max_host_gap_ms": round(max(host) * 1e3, 3),
                     "sum_host_ms": round(sum(host) * 1e3, 2), "gc_generations": gcs})
        print(json.dumps(rows[-1]), flush=True)
    s = sorted(r["single_pair_ms_per_call"] for r in rows)
    m = sorted(r["per_call_median_ms"] for r in rows)
    print(json.dumps({"summary": {"windows": len(rows), "single_pair": {"min": s[0], "median": s[len(s) // 2], "max": s[-1]},
                                  "per_call_median": {"min": m[0], "median": m[len(m) // 2], "max": m[-1]},
                                  "windows_with_gc": sum(1 for r in rows if r["gc_generations"]),
                                  "worst_host_gap_ms": max(r["max_host_gap_ms"] for r in rows)}}), flush=True)


if __name__ == "__main__":
    main()

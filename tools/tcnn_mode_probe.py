"""Why is the tcnn-layout / fp16-table render (the mode an imported reference checkpoint runs in) bimodal between runs?

Per-LAUNCH HIP-event times (min / median / max / sigma) of one C2 batch (65 536 rays x 192 samples) through cn_render_rays
for the headline table and for the tcnn fp16 table under controlled conditions:

  * the table freshly packed (what bench.py did in round 2), the same table cloned into new allocations (placement),
    the table inside one early 1 GiB arena, after churn of the caching allocator;
  * right after 20 s of an idle GPU (the CPU baseline of bench.py leaves the GPU idle for ~35 s before the secondaries);
  * one warm-up launch (round 2's bench) against ten.

    python tools/tcnn_mode_probe.py [--launches 60] [--quick]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def per_launch(ops, fh, scene, batches, n, warm, **opt_kw):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]

    def one(i):
        o, d, nn, f, cam, start = batches[i % len(batches)]
        ops.render_rays(fh, scene, ops.render_opts(bench.S, image_width=bench.W, pixel_start=start, **opt_kw), o, d, nn, f)

    for i in range(warm):
        one(i)
    for i in range(n):
        evs[i][0].record()
        one(i)
        evs[i][1].record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in evs]
    per_batch = [round(statistics.median(t[b::len(batches)]), 3) for b in range(min(len(batches), n))]
    return {"n": n, "warm": warm, "min": round(min(t), 3), "median": round(statistics.median(t), 3),
            "mean": round(sum(t) / n, 3), "max": round(max(t), 3), "sigma": round(statistics.pstdev(t), 3),
            "first3": [round(v, 3) for v in t[:3]], "per_batch_median": per_batch}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=60)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    from cropnerf_amd import _lib as L
    from cropnerf_amd import config, ops

    dev = torch.device("cuda", 0)
    arena = torch.empty(1 << 30, dtype=torch.uint8, device=dev)  # one early 1 GiB allocation
    cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
    batches = bench.make_batches(ops, c2w, intr, 0, 1)
    scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)
    n = args.launches

    def show(tag, res):
        print(json.dumps({"case": tag, **res}), flush=True)

    show("headline torch_f32", per_launch(ops, fh, scene, batches, n, 5))
    tcfg = config.FruitNerfModelConfig(num_nerf_samples_per_ray=bench.S, implementation="tcnn")
    tf = tcfg.field_spec(num_images=bench.NUM_CAMERAS)
    g = torch.Generator(device="cpu").manual_seed(0)
    packed = ((torch.rand(2 * tf.grid.num_packed_entries, generator=g) * 2 - 1) * 0.1).to(dev)

    def handle(table):
        pt = dict(params)
        pt["field.mlp_base_grid.hash_table"] = table
        return ops.FieldHandle(pt, tf)

    t16 = ops.tcnn_grid_pack(tf.grid, packed, torch.float16)
    show("tcnn_f16 fresh pack, 1 warm-up (round-2 bench)", per_launch(ops, handle(t16), scene, batches, 20, 1))
    show("tcnn_f16 fresh pack, again", per_launch(ops, handle(t16), scene, batches, n, 5))
    print(json.dumps({"table_ptr": hex(t16.data_ptr()), "bytes": t16.numel() * 2, "mod_2MiB": t16.data_ptr() % (1 << 21)}), flush=True)
    # placement: the same values in other allocations
    keep = []
    for k in range(3 if args.quick else 6):
        keep.append(torch.empty((3 + 5 * k) << 20, dtype=torch.uint8, device=dev))  # shift what the allocator hands out next
        c = t16.clone()
        keep.append(c)
        r = per_launch(ops, handle(c), scene, batches, n, 5)
        r["ptr"] = hex(c.data_ptr())
        show(f"tcnn_f16 clone {k}", r)
    # inside the early arena, 2 MiB aligned
    base = (arena.data_ptr() + (1 << 21) - 1) // (1 << 21) * (1 << 21) - arena.data_ptr()
    nb = t16.numel() * 2
    a16 = arena[base:base + nb].view(torch.float16).view(t16.shape)
    a16.copy_(t16)
    show("tcnn_f16 in the early 1 GiB arena", per_launch(ops, handle(a16), scene, batches, n, 5))
    a16b = arena[base + (512 << 20) + 4096:base + (512 << 20) + 4096 + nb].view(torch.float16).view(t16.shape)
    a16b.copy_(t16)
    show("tcnn_f16 in the arena at +512 MiB + 4 KiB", per_launch(ops, handle(a16b), scene, batches, n, 5))
    # the other three table kinds for reference
    t32 = ops.tcnn_grid_pack(tf.grid, packed, torch.float32)
    show("tcnn_f32", per_launch(ops, handle(t32), scene, batches, n, 5))
    ph = dict(params)
    ph["field.mlp_base_grid.hash_table"] = params["field.mlp_base_grid.hash_table"].to(torch.float16)
    show("torch_f16", per_launch(ops, ops.FieldHandle(ph, fspec), scene, batches, n, 5))
    show("headline torch_f32 again", per_launch(ops, fh, scene, batches, n, 5))
    # after an idle GPU (bench.py's CPU baseline)
    if not args.quick:
        torch.cuda.synchronize()
        time.sleep(20.0)
        show("tcnn_f16 after 20 s idle, 1 warm-up", per_launch(ops, handle(t16), scene, batches, 20, 1))
        time.sleep(20.0)
        show("headline after 20 s idle, 1 warm-up", per_launch(ops, fh, scene, batches, 20, 1))
        # busy host: 16 torch threads grinding while the GPU renders (bench.py's CPU baseline never overlaps, but a loaded
        # host delays launches)
        show("tcnn_f16 steady", per_launch(ops, handle(t16), scene, batches, n, 10))
    for mp, name in ((L.MATRIX_SPLIT_BF16, "split_bf16"),):
        show(f"tcnn_f16 {name}", per_launch(ops, handle(t16), scene, batches, n, 5, matrix_precision=mp))
        show(f"torch_f32 {name}", per_launch(ops, fh, scene, batches, n, 5, matrix_precision=mp))


if __name__ == "__main__":
    main()

"""Timing probe: one C2 batch (65 536 rays x 192 samples, uniform sampler) through cn_render_rays for the grid layouts /
table types / matrix precisions.  HIP-event average over the distinct batches of bench.py.

    python tools/render_ab.py [--iters 3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_mode(ops, fh, scene, batches, iters, **opt_kw):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    def run():
        for (o, d, n, f, cam, start) in batches:
            ops.render_rays(fh, scene, ops.render_opts(bench.S, image_width=bench.W, pixel_start=start, **opt_kw), o, d, n, f)
    run()
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / (iters * len(batches))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    from cropnerf_amd import _lib as L
    from cropnerf_amd import config, ops, synthetic
    from cropnerf_amd.fruit_nerf import tcnn_params

    dev = torch.device("cuda", 0)
    cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
    batches = bench.make_batches(ops, c2w, intr, 0, 1)
    scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)
    res = {}
    res["torch_f32"] = time_mode(ops, fh, scene, batches, args.iters)
    res["torch_f32_bf16mm"] = time_mode(ops, fh, scene, batches, args.iters, matrix_precision=L.MATRIX_SPLIT_BF16)
    res["torch_f32_f16mm"] = time_mode(ops, fh, scene, batches, args.iters, matrix_precision=L.MATRIX_F16)
    ph = dict(params)
    ph["field.mlp_base_grid.hash_table"] = params["field.mlp_base_grid.hash_table"].to(torch.float16)
    fhh = ops.FieldHandle(ph, fspec)
    res["torch_f16"] = time_mode(ops, fhh, scene, batches, args.iters)
    res["torch_f16_bf16mm"] = time_mode(ops, fhh, scene, batches, args.iters, matrix_precision=L.MATRIX_SPLIT_BF16)
    res["torch_f16_f16mm"] = time_mode(ops, fhh, scene, batches, args.iters, matrix_precision=L.MATRIX_F16)
    # tcnn layout: same MLPs, a random table of that layout
    tcfg = config.FruitNerfModelConfig(num_nerf_samples_per_ray=bench.S, implementation="tcnn")
    tf = tcfg.field_spec(num_images=bench.NUM_CAMERAS)
    g = torch.Generator(device="cpu").manual_seed(0)
    packed = ((torch.rand(2 * tf.grid.num_packed_entries, generator=g) * 2 - 1) * 0.1).to(dev)
    for name, dt in (("tcnn_f32", torch.float32), ("tcnn_f16", torch.float16)):
        pt = dict(params)
        pt["field.mlp_base_grid.hash_table"] = ops.tcnn_grid_pack(tf.grid, packed, dt)
        fht = ops.FieldHandle(pt, tf)
        res[name] = time_mode(ops, fht, scene, batches, args.iters)
        res[name + "_bf16mm"] = time_mode(ops, fht, scene, batches, args.iters, matrix_precision=L.MATRIX_SPLIT_BF16)
        res[name + "_f16mm"] = time_mode(ops, fht, scene, batches, args.iters, matrix_precision=L.MATRIX_F16)
    for k, v in res.items():
        print(f"{k:22s} {v:7.3f} ms/batch  {bench.R * bench.S / v / 1e6:8.1f} Msamples/s", flush=True)


if __name__ == "__main__":
    main()

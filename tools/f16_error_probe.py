"""Error statistics of the fp16 matrix mode (CN_MATRIX_F16) on the small tcnn test scene: HIP render against the oracle with
tcnn's own activation rounding (tcnn_half_activations=True), against the fp32-arithmetic oracle, and against the HIP fp32
render -- the numbers the tolerances of tests/test_gpu_f16.py are set from.

    python tools/f16_error_probe.py
"""
import dataclasses
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _helpers import dev_params, make_tcnn_scene, make_scene, oracle_model, product_specs, rays_with_box, to_dev  # noqa: E402


def stats(name, a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    e = (a - b).abs()
    mse = float(((a - b) ** 2).mean())
    print(f"  {name:34s} max {e.max().item():.3e}  mean {e.mean().item():.3e}  p99 {e.flatten().quantile(0.99).item():.3e}"
          f"  rel-L2 {float((a - b).norm() / b.norm().clamp_min(1e-12)):.3e}  psnr {(-10 * math.log10(max(mse, 1e-30))):.1f} dB")


def main():
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops

    for grid_scale in (0.1, 1.0):
        for S, contraction in ((192, False), (64, True)):
            scene = make_tcnn_scene(seed=3, grid_scale=grid_scale)
            fspec, _ = product_specs(scene)
            fh = ops.FieldHandle(dev_params(scene), fspec)
            rb = rays_with_box(scene, 0, 800)
            m32 = oracle_model(scene, "inference", disable_scene_contraction=not contraction)
            m32.uniform_samples = S
            ref32 = m32.forward(rb)
            sc16 = dataclasses.replace(scene, fspec=dataclasses.replace(scene.fspec, tcnn_half_activations=True))
            m16 = oracle_model(sc16, "inference", disable_scene_contraction=not contraction)
            m16.uniform_samples = S
            ref16 = m16.forward(rb)
            scn = ops.scene_struct(scene.aabb, contraction)
            args = [to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars)]
            os.environ["CN_FUSED_SPLIT"] = "2"
            out32 = ops.render_rays(fh, scn, ops.render_opts(S), *args, want_weights=True)
            out16 = ops.render_rays(fh, scn, ops.render_opts(S, matrix_precision=L.MATRIX_F16), *args, want_weights=True)
            print(f"grid_scale {grid_scale} S {S} contraction {contraction}")
            for k, rk in (("rgb", "rgb"), ("accumulation", "accumulation"), ("semantics", "semantics"), ("weights", "_weights")):
                r16 = ref16[rk][..., 0] if rk == "_weights" else ref16[rk]
                r32 = ref32[rk][..., 0] if rk == "_weights" else ref32[rk]
                stats(f"{k}: f16 HIP vs oracle(half act)", out16[k], r16)
                stats(f"{k}: f16 HIP vs oracle(fp32)", out16[k], r32)
                stats(f"{k}: oracle(half) vs oracle(fp32)", r16, r32)
                stats(f"{k}: f16 HIP vs fp32 HIP", out16[k], out32[k])
            d = (out16["depth"].cpu() - ref16["depth"]).abs()
            print(f"  depth equal frac (1e-5): {(d <= 1e-5 + 1e-5 * ref16['depth'].abs()).float().mean().item():.4f}")
    # torch-layout fp32 model through the fp16 mode (weights rounded to fp16 on the fly)
    sc_ = make_scene(seed=5, log2_T=15, prop_log2_T=12)
    fspec, _ = product_specs(sc_)
    fh = ops.FieldHandle(dev_params(sc_), fspec)
    rb = rays_with_box(sc_, 0, 800)
    args = [to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars)]
    scn = ops.scene_struct(sc_.aabb, False)
    out32 = ops.render_rays(fh, scn, ops.render_opts(96), *args)
    out16 = ops.render_rays(fh, scn, ops.render_opts(96, matrix_precision=L.MATRIX_F16), *args)
    print("torch layout, fp32 table")
    for k in ("rgb", "accumulation", "semantics"):
        stats(f"{k}: f16 HIP vs fp32 HIP", out16[k], out32[k])


if __name__ == "__main__":
    main()

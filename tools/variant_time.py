"""Time one C2 batch through cn_render_rays for a few table / matrix modes with whatever library CROPNERF_HIP_LIB names
(timing-only variant builds: tools/build_variant.sh).   python tools/variant_time.py [modes...]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops

    want = sys.argv[1:] or ["torch_f32", "torch_f32_f16mm", "tcnn_f16", "tcnn_f16_bf16mm", "tcnn_f16_f16mm"]
    dev = torch.device("cuda", 0)
    cfg, fspec, pspecs, params, fh, dh, c2w, intr = bench.build_scene(dev)
    batches = bench.make_batches(ops, c2w, intr, 0, 1)
    scene = ops.scene_struct(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), contraction=False)
    fht, _ = bench.tcnn_f16_field(params, dev)
    mm = {"": L.MATRIX_FP32, "_bf16mm": L.MATRIX_SPLIT_BF16, "_f16mm": L.MATRIX_F16}
    for name in want:
        base = name.replace("_bf16mm", "").replace("_f16mm", "")
        h = fh if base == "torch_f32" else fht
        kw = {"matrix_precision": mm[name[len(base):]]}
        n = 40
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n + 5):
            o, d, nn, f, cam, start = batches[i % len(batches)]
            if i >= 5:
                evs[i - 5][0].record()
            ops.render_rays(h, scene, ops.render_opts(bench.S, image_width=bench.W, pixel_start=start, **kw), o, d, nn, f)
            if i >= 5:
                evs[i - 5][1].record()
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in evs]
        print(f"{os.path.basename(os.environ.get('CROPNERF_HIP_LIB', 'default')):34s} {name:18s} median {statistics.median(t):6.3f}  "
              f"mean {sum(t) / n:6.3f}  min {min(t):6.3f}  max {max(t):6.3f} ms", flush=True)


if __name__ == "__main__":
    main()

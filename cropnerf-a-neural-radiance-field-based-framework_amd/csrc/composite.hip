// cn_composite: compositing of materialised per-sample field outputs, one wavefront per ray.
// HBM-streaming: reads (2 + 1 + 3 + 1) * 4 B per sample, writes ~28 B per ray (+ 4 B/sample when weights are kept).
#include "composite_dev.hpp"

namespace cn {

__global__ void __launch_bounds__(256)
composite_kernel(const float* __restrict__ starts, const float* __restrict__ ends, const float* __restrict__ density,
                 const float* __restrict__ rgb, const float* __restrict__ sem, long long num_rays, int S, int bg_mode,
                 float bgr, float bgg, float bgb, int eval_clamp, float* __restrict__ out_rgb,
                 float* __restrict__ out_acc, float* __restrict__ out_depth, float* __restrict__ out_sem,
                 float* __restrict__ out_cmap, float* __restrict__ out_w) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = blockIdx.x * (long long)(blockDim.x >> 6) + wave; r < num_rays; r += waves) {
    CompositeState st;
    const long long base = r * (long long)S;
    for (int c0 = 0; c0 < S; c0 += 64) {
      int i = c0 + lane;
      bool valid = i < S;
      int ic = valid ? i : S - 1;
      float s0 = starts[base + ic], e0 = ends[base + ic];
      float den = density[base + ic];
      float cr = 0.f, cg = 0.f, cb = 0.f, sm = 0.f;
      if (rgb) {
        cr = rgb[3 * (base + ic) + 0];
        cg = rgb[3 * (base + ic) + 1];
        cb = rgb[3 * (base + ic) + 2];
      }
      if (sem) sm = sem[base + ic];
      float w = composite_chunk(st, valid, i == S - 1, e0 - s0, den, (s0 + e0) / 2.f, cr, cg, cb, sm, eval_clamp != 0);
      if (out_w && valid) out_w[base + i] = w;
    }
    CompositeOut o = composite_finish(st, bg_mode, bgr, bgg, bgb, eval_clamp != 0);
    if (lane == 0) {
      if (out_rgb && rgb) {
        out_rgb[3 * r + 0] = o.r;
        out_rgb[3 * r + 1] = o.g;
        out_rgb[3 * r + 2] = o.b;
      }
      if (out_acc) out_acc[r] = o.acc;
      if (out_depth) out_depth[r] = o.depth;
      if (out_sem && sem) out_sem[r] = o.sem;
      if (out_cmap && sem) {
        float l = semantics_label(o.sem);
        out_cmap[3 * r + 0] = l;
        out_cmap[3 * r + 1] = l;
        out_cmap[3 * r + 2] = l;
      }
    }
  }
}

}  // namespace cn

extern "C" int cn_composite(const float* starts, const float* ends, const float* density, const float* rgb,
                            const float* semantics, int64_t num_rays, int32_t num_samples, int32_t bg_mode,
                            const float* bg_color, int32_t eval_clamp, float* out_rgb, float* out_accumulation,
                            float* out_depth, float* out_semantics, float* out_semantics_colormap, float* out_weights,
                            cn_stream_t stream) {
  CN_REQUIRE(starts && ends && density, CN_ERR_INVALID, "cn_composite: null starts/ends/density");
  CN_REQUIRE(num_samples > 0, CN_ERR_INVALID, "cn_composite: num_samples must be > 0");
  CN_REQUIRE(bg_mode == CN_BG_LAST_SAMPLE || (bg_mode == CN_BG_COLOR && bg_color), CN_ERR_INVALID,
             "cn_composite: bad background mode %d", bg_mode);
  if (num_rays <= 0) return CN_OK;
  float b0 = bg_color ? bg_color[0] : 0.f, b1 = bg_color ? bg_color[1] : 0.f, b2 = bg_color ? bg_color[2] : 0.f;
  hipLaunchKernelGGL(cn::composite_kernel, dim3(cn::grid_for(num_rays, 4, 16384)), dim3(256), 0,
                     cn::as_stream(stream), starts, ends, density, rgb, semantics, (long long)num_rays, num_samples,
                     bg_mode, b0, b1, b2, eval_clamp, out_rgb, out_accumulation, out_depth, out_semantics,
                     out_semantics_colormap, out_weights);
  return cn::check_launch("cn_composite");
}

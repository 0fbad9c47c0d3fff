// Wave-level alpha compositing: one wavefront owns one ray and consumes its samples 64 at a time in lane order.
// Restates RaySamples.get_weights + RGB / Accumulation / Depth(median) / Semantic renderers
// (fruit_nerf/fruit_nerf.py:556-597).
#pragma once

#include "cn_common.hpp"
#include "wave_ops.hpp"

namespace cn {

struct CompositeState {
  float carry_dd = 0.f;  // sum of delta*density over earlier chunks (wave-uniform)
  float carry_w = 0.f;   // sum of weights over earlier chunks (wave-uniform)
  float r = 0.f, g = 0.f, b = 0.f, w = 0.f, s = 0.f;  // per-lane partial sums
  float depth = 0.f;  // median depth once found (wave-uniform)
  int found = 0;
  float last_r = 0.f, last_g = 0.f, last_b = 0.f, last_mid = 0.f;  // sample S-1 (wave-uniform)
};

// One chunk of <=64 samples, lane l = sample chunk_base + l.  Returns this lane's weight.
__device__ __forceinline__ float composite_chunk(CompositeState& st, bool valid, bool is_last_sample, float delta,
                                                 float density, float mid, float cr, float cg, float cb, float sem,
                                                 bool eval_clamp) {
  float dd = valid ? delta * density : 0.f;
  float incl = wave_inclusive_scan(dd);
  float trans = expf(-(st.carry_dd + (incl - dd)));
  float alpha = 1.f - expf(-dd);
  float wgt = valid ? nan_to_num(alpha * trans) : 0.f;
  st.carry_dd += wave_read(incl, 63);
  if (eval_clamp) {
    cr = nan_to_num(cr);
    cg = nan_to_num(cg);
    cb = nan_to_num(cb);
  }
  st.r += wgt * cr;
  st.g += wgt * cg;
  st.b += wgt * cb;
  st.w += wgt;
  st.s += wgt * sem;
  float cw = st.carry_w + wave_inclusive_scan(wgt);
  unsigned long long hit = __ballot(valid && cw >= 0.5f);
  if (!st.found && hit) {
    int src = __ffsll((long long)hit) - 1;
    st.depth = wave_read(mid, src);
    st.found = 1;
  }
  st.carry_w = wave_read(cw, 63);
  unsigned long long last = __ballot(valid && is_last_sample);
  if (last) {
    int src = __ffsll((long long)last) - 1;
    st.last_r = wave_read(cr, src);
    st.last_g = wave_read(cg, src);
    st.last_b = wave_read(cb, src);
    st.last_mid = wave_read(mid, src);
  }
  return wgt;
}

struct CompositeOut {
  float r, g, b, acc, depth, sem;
};

__device__ __forceinline__ CompositeOut composite_finish(const CompositeState& st, int bg_mode, float bgr, float bgg,
                                                         float bgb, bool eval_clamp) {
  CompositeOut o;
  float acc = wave_sum(st.w);
  float r = wave_sum(st.r), g = wave_sum(st.g), b = wave_sum(st.b);
  o.sem = wave_sum(st.s);
  if (bg_mode == CN_BG_LAST_SAMPLE) {
    bgr = st.last_r;
    bgg = st.last_g;
    bgb = st.last_b;
  }
  float k = 1.f - acc;
  r += bgr * k;
  g += bgg * k;
  b += bgb * k;
  if (eval_clamp) {
    r = fminf(fmaxf(r, 0.f), 1.f);
    g = fminf(fmaxf(g, 0.f), 1.f);
    b = fminf(fmaxf(b, 0.f), 1.f);
  }
  o.r = r;
  o.g = g;
  o.b = b;
  o.acc = acc;
  o.depth = st.found ? st.depth : st.last_mid;  // clamp(searchsorted, 0, S-1)
  return o;
}

// heaviside(sigmoid(sem) - 0.9, 0) -> {0,1}  (fruit_nerf.py:593-597; colormap = (0,1))
__device__ __forceinline__ float semantics_label(float sem) { return (sigmoidf(sem) - 0.9f) > 0.f ? 1.f : 0.f; }

}  // namespace cn

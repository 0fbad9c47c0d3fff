// Wave-cooperative sampler building blocks (one wavefront owns one ray; per-ray arrays live in LDS).
// Restates nerfstudio's PDFSampler.generate_ray_samples / RaySamples.get_weights as used through
// ProposalNetworkSampler at fruit_nerf/fruit_nerf.py:157-164,549.
#pragma once

#include "cn_common.hpp"
#include "wave_ops.hpp"

namespace cn {

// cdf[0..s_in] from weights w[0..s_in): w' = w^anneal + 0.01; pad; pdf = w'/sum; cdf = min(1, cumsum).
// Every lane must call; LDS arrays are per wave.  Ends with the data visible to the whole wave.
__device__ __forceinline__ void wave_cdf_from_weights(const float* w, int s_in, float anneal, float* cdf) {
  const int lane = lane_id();
  const int per = (s_in + 63) >> 6;
  const int e0 = lane * per;
  const int e1 = min(e0 + per, s_in);
  const float hist_pad = 0.01f, eps = 1e-5f;
  float local = 0.f;
  for (int e = e0; e < e1; ++e) {
    float v = w[e];
    if (anneal != 1.f) v = powf(v, anneal);
    local += v + hist_pad;
  }
  float total = wave_sum(local);
  float padding = fmaxf(eps - total, 0.f);
  float add = padding / (float)s_in;
  total += padding;
  // second pass: pdf and its running sum
  float lsum = 0.f;
  for (int e = e0; e < e1; ++e) {
    float v = w[e];
    if (anneal != 1.f) v = powf(v, anneal);
    lsum += (v + hist_pad + add) / total;
  }
  float incl = wave_inclusive_scan(lsum);
  float run = incl - lsum;
  for (int e = e0; e < e1; ++e) {
    float v = w[e];
    if (anneal != 1.f) v = powf(v, anneal);
    run += (v + hist_pad + add) / total;
    cdf[e + 1] = fminf(1.f, run);
  }
  if (lane == 0) cdf[0] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// u value of output bin b (PDFSampler: linspace(0, 1-1/nb, nb) + 1/(2nb) in eval, + rand/nb in training)
__device__ __forceinline__ float pdf_u(int b, int nb, const float* u_rand_row, int u_rand_stride) {
  float end = (float)(1.0 - 1.0 / (double)nb);
  float step = nb > 1 ? end / (float)(nb - 1) : 0.f;
  float u = b < nb / 2 ? step * (float)b : end - step * (float)(nb - b - 1);
  if (u_rand_row) {
    float r = u_rand_stride == 1 ? u_rand_row[0] : u_rand_row[b];
    return u + r / (float)nb;
  }
  return u + (float)(1.0 / (2.0 * (double)nb));
}

// Invert the cdf at u: searchsorted(right), gather below/above, lerp in the spacing domain.
__device__ __forceinline__ float pdf_invert(const float* cdf, const float* prev_bins, int s_in, float u) {
  // first index i in [0, s_in+1) with cdf[i] > u   (torch.searchsorted side="right")
  int lo = 0, hi = s_in + 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  int below = min(max(lo - 1, 0), s_in);
  int above = min(max(lo, 0), s_in);
  float c0 = cdf[below], c1 = cdf[above];
  float b0 = prev_bins[below], b1 = prev_bins[above];
  float t = (u - c0) / (c1 - c0);
  t = nan_to_num(t);
  t = fminf(fmaxf(t, 0.f), 1.f);
  return b0 + t * (b1 - b0);
}

}  // namespace cn

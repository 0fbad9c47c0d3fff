// Materialising samplers (API completeness + parity tests; the fused renderer samples in registers).
//   cn_sample_spaced : fruit_nerf/components/ray_samplers.py:54-104 (UniformSamplerWithNoise / SpacedSampler)
//   cn_sample_pdf    : nerfstudio PDFSampler via ProposalNetworkSampler (fruit_nerf/fruit_nerf.py:157-164)
#include "sampler_dev.hpp"

namespace cn {

__global__ void __launch_bounds__(256)
sample_spaced_kernel(const float* __restrict__ nears, const float* __restrict__ fars, long long num_rays, int S,
                     int spacing, const float* __restrict__ t_rand, int t_stride, float* __restrict__ starts,
                     float* __restrict__ ends, float* __restrict__ sp_starts, float* __restrict__ sp_ends) {
  long long total = num_rays * (long long)S;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i / S;
    int k = (int)(i - r * S);
    float b0 = linspace01(k, S + 1), b1 = linspace01(k + 1, S + 1);
    if (t_rand) {
      // bins = lower + (upper-lower)*rand with lower/upper from the bin centres (ray_samplers.py:84-87)
      auto jitter = [&](int e) {
        float be = linspace01(e, S + 1);
        float lo = e == 0 ? be : (be + linspace01(e - 1, S + 1)) / 2.f;
        float hi = e == S ? be : (linspace01(e + 1, S + 1) + be) / 2.f;
        float t = t_stride == 1 ? t_rand[r] : t_rand[r * t_stride + e];
        return lo + (hi - lo) * t;
      };
      b0 = jitter(k);
      b1 = jitter(k + 1);
    }
    float sn = spacing_fn(spacing, nears[r]), sf = spacing_fn(spacing, fars[r]);
    if (starts) starts[i] = spacing_to_euclid(spacing, b0, sn, sf);
    if (ends) ends[i] = spacing_to_euclid(spacing, b1, sn, sf);
    if (sp_starts) sp_starts[i] = b0;
    if (sp_ends) sp_ends[i] = b1;
  }
}

// one wave per ray; LDS per wave: prev bins [s_in+1] | weights [s_in] | cdf [s_in+1]
__global__ void __launch_bounds__(256)
sample_pdf_kernel(const float* __restrict__ prev_bins, const float* __restrict__ weights,
                  const float* __restrict__ nears, const float* __restrict__ fars, long long num_rays, int s_in,
                  int s_out, float anneal, int spacing, const float* __restrict__ u_rand, int u_stride,
                  float* __restrict__ out_sp, float* __restrict__ out_eu) {
  extern __shared__ __align__(16) float lds[];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int per_wave = 3 * s_in + 2;
  float* pb = lds + wave * per_wave;
  float* w = pb + s_in + 1;
  float* cdf = w + s_in;
  const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = blockIdx.x * (long long)(blockDim.x >> 6) + wave; r < num_rays; r += waves) {
    for (int e = lane; e <= s_in; e += 64) pb[e] = prev_bins[r * (s_in + 1) + e];
    for (int e = lane; e < s_in; e += 64) w[e] = weights[r * s_in + e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    wave_cdf_from_weights(w, s_in, anneal, cdf);
    const int nb = s_out + 1;
    float sn = spacing_fn(spacing, nears[r]), sf = spacing_fn(spacing, fars[r]);
    const float* ur = u_rand ? u_rand + r * u_stride : nullptr;
    for (int b = lane; b < nb; b += 64) {
      float u = pdf_u(b, nb, ur, u_stride);
      float bin = pdf_invert(cdf, pb, s_in, u);
      if (out_sp) out_sp[r * nb + b] = bin;
      if (out_eu) out_eu[r * nb + b] = spacing_to_euclid(spacing, bin, sn, sf);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace cn

extern "C" int cn_sample_spaced(const float* nears, const float* fars, int64_t num_rays, int32_t num_samples,
                                int32_t spacing, const float* t_rand, int32_t t_rand_stride, float* starts,
                                float* ends, float* spacing_starts, float* spacing_ends, cn_stream_t stream) {
  CN_REQUIRE(nears && fars, CN_ERR_INVALID, "cn_sample_spaced: null nears/fars");
  CN_REQUIRE(num_samples > 0, CN_ERR_INVALID, "cn_sample_spaced: num_samples must be > 0");
  CN_REQUIRE(spacing == CN_SPACING_UNIFORM || spacing == CN_SPACING_PIECEWISE, CN_ERR_INVALID,
             "cn_sample_spaced: unknown spacing %d", spacing);
  CN_REQUIRE(!t_rand || t_rand_stride == 1 || t_rand_stride == num_samples + 1, CN_ERR_INVALID,
             "cn_sample_spaced: t_rand_stride must be 1 or S+1");
  if (num_rays <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::sample_spaced_kernel, dim3(cn::grid_for(num_rays * num_samples, 256, 16384)), dim3(256), 0,
                     cn::as_stream(stream), nears, fars, (long long)num_rays, num_samples, spacing, t_rand,
                     t_rand_stride, starts, ends, spacing_starts, spacing_ends);
  return cn::check_launch("cn_sample_spaced");
}

extern "C" int cn_sample_pdf(const float* prev_spacing_bins, const float* weights, const float* nears,
                             const float* fars, int64_t num_rays, int32_t s_in, int32_t s_out, float anneal,
                             int32_t spacing, const float* u_rand, int32_t u_rand_stride, float* spacing_bins,
                             float* euclidean_bins, cn_stream_t stream) {
  CN_REQUIRE(prev_spacing_bins && weights && nears && fars, CN_ERR_INVALID, "cn_sample_pdf: null input");
  CN_REQUIRE(s_in > 0 && s_in <= 4096 && s_out > 0, CN_ERR_INVALID, "cn_sample_pdf: bad sample counts %d -> %d", s_in,
             s_out);
  CN_REQUIRE(!u_rand || u_rand_stride == 1 || u_rand_stride == s_out + 1, CN_ERR_INVALID,
             "cn_sample_pdf: u_rand_stride must be 1 or S_out+1");
  if (num_rays <= 0) return CN_OK;
  size_t lds = (size_t)4 * (3 * s_in + 2) * sizeof(float);
  hipLaunchKernelGGL(cn::sample_pdf_kernel, dim3(cn::grid_for(num_rays, 4, 4096)), dim3(256), lds,
                     cn::as_stream(stream), prev_spacing_bins, weights, nears, fars, (long long)num_rays, s_in, s_out,
                     anneal, spacing, u_rand, u_rand_stride, spacing_bins, euclidean_bins);
  return cn::check_launch("cn_sample_pdf");
}

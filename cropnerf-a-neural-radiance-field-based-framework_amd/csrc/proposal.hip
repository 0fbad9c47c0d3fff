// cn_proposal_sample: the whole ProposalNetworkSampler of fruit_nerf (fruit_nerf/fruit_nerf.py:157-164, called at
// :549,501,429,337) in one launch, eval mode: piecewise-in-disparity initial bins -> proposal net 0 -> weights ->
// PDF resample -> proposal net 1 -> weights -> PDF resample -> final bins.  One wavefront per ray; the per-ray
// bins / weights / cdf live in LDS, nothing [R,S]-shaped goes to HBM except the final bins ((S+1)*4 B per ray) and
// the two median depths.  Proposal nets are small (L levels x 8 gathers, 2L->16->1): plain VALU with the MLP
// weights in wave-uniform (scalar) registers.
//
// Built for the proposal shapes the reference configs use: 5 or 7 levels, hidden 16
// (nerfacto defaults / fruit_nerf_config.py:143-146); other shapes return CN_ERR_UNSUPPORTED and the caller
// composes cn_sample_spaced + cn_proposal_density + cn_composite + cn_sample_pdf instead.
#include <algorithm>

#include "composite_dev.hpp"
#include "sampler_dev.hpp"

namespace cn {

// What bounds this kernel, and what was tried (round 3; C2 batch, eval):
//  * x-pair gathers (hash_level_xpair: 40 gathers per evaluation become 30, 5.8e8 L1 line lookups per launch) bought 2 % while the
//    MLP ran on the VALU (1.054 -> 1.030 ms) and LOSE now that it does not (0.678 vs 0.609 ms with plain hash_level): the selects and
//    the branch of the pair form are VALU instructions, and VALU issue is what is left.  CN_PROP_XPAIR=1 builds it.
//  * the team form of the render kernels -- the four waves of a workgroup (four consecutive rays, neighbouring pixels) each
//    evaluating 16 samples of a chunk for all four rays, rays in adjacent lanes, densities handed over through LDS; bit-identical
//    results -- left the launch at 1.028 ms: the L1 lookups never bounded it.  Removed.
//  * one round trip per level instead of two (pair loads and odd-x loads issued back to back): 1.043 ms; levels software-pipelined:
//    0.85 vs 0.67 ms (spills); five waves per SIMD: 0.70 vs 0.67 ms (spills).
//  * the sampler scaffolding (0.21 of the 0.61 ms): linspace step and the pdf_u constants hoisted out of their loops by hand and the
//    inverse-cdf search written without data-dependent control flow (bit-identical results): 0.616 vs 0.608 ms -- hipcc had hoisted
//    what is invariant, and ten uniform steps cost what nine divergent ones do.  Not kept.
//  * where the instructions are now (SQ_INSTS_VALU per ray, profiles/r03e_pmc_proposal_sampler.json and ablation builds of the
//    scaffolding): 4 213 in all -- the two networks 2 210 (352 evaluations x ~362 / 64 lanes), the scaffolding 2 003 = the two cdfs 217
//    + their inversions 497 + the compositing scans 375 + bins, positions, spacing functions and outputs 877; 84 % of the SIMDs'
//    issue slots are taken.
//  * ablation builds (-DCN_PROP_ABLATE=1 / 2), MLP on the VALU: 0.21 ms without any network, 0.58 ms with the hash encoding, 1.04 ms
//    complete -- the MLP was the larger half; see prop_mlp_mfma.
#ifndef CN_PROP_XPAIR
#define CN_PROP_XPAIR 0
#endif
#ifndef CN_PROP_ABLATE
#define CN_PROP_ABLATE 0
#endif
constexpr int PROP_MAX_LEVELS = 3;   // proposal iterations
constexpr int PROP_MAX_SAMPLES = 512;

struct PropNet {
  GridDev grid;
  const float *w0, *b0, *w1, *b1;
};

struct PropArgs {
  PropNet net[PROP_MAX_LEVELS];
  int num_levels;
  int s_prop[PROP_MAX_LEVELS];
  int s_final;
  int smax;
  int enc_rows;  // rows of a wave's encoding block (prop_mlp_mfma): 2 L rounded up to a multiple of 4, of the largest net
  float anneal;
  const float* anneal_dev;  // when not NULL the exponent is read from device memory (a captured HIP graph replays with new values)
  SceneDev scene;
  const float* origins;
  const float* directions;
  const float* nears;
  const float* fars;
  long long num_rays;
  float* out_eu;
  float* out_sp;
  float* out_depth;  // [num_levels][R]
  // training (TRAIN = true): one uniform random per ray and resampling step, and what the interlevel loss and the
  // proposal backward need of every level
  const float* jitter;                 // [num_levels + 1][R]
  float* lv_sp[PROP_MAX_LEVELS];       // [R, S_l + 1] spacing bins
  float* lv_starts[PROP_MAX_LEVELS];   // [R, S_l]
  float* lv_ends[PROP_MAX_LEVELS];     // [R, S_l]
  float* lv_density[PROP_MAX_LEVELS];  // [R, S_l]
  float *final_starts, *final_ends;    // [R, s_final] (optional): euclidean_bins[:, :-1] / [:, 1:] as contiguous arrays
};

// The 2L -> 16 -> 1 MLP of a proposal network on the fp32 matrix cores (CN_PROP_MLP_MFMA, default).  On the VALU it was the
// larger half of the kernel (ablation builds, C2 batch: 0.21 ms without any network, 0.58 ms with the hash encoding, 1.04 ms
// with the MLP: 176 multiply-adds per evaluation, whose 193 weights do not fit the scalar registers and were re-fetched with
// scalar loads for every 64 evaluations).  Here the wave's 64 evaluations are four 16-column tiles of one product
// hid[16 x 64] = W0[16 x K] enc[K x 64]: W0 (K padded to a multiple of 4 with zero columns) is the A operand, held in K / 4
// registers per lane for a whole level; the encodings change lanes through a wave-private LDS block [k][sample] (row stride 80:
// the four k-rows a B operand reads fall into different banks); the accumulators start from the bias, and the second layer is four
// multiply-adds per lane and tile plus a three-step butterfly over the four lane groups that leaves evaluation l in lane l.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PROP_ENC_STRIDE = 80;  // per wave: [enc_rows][80] floats, enc_rows = 12 (5-level nets) or 16 (7 levels)
struct PropMlp {
  float a[4];   // W0[m = lane & 15][k = 4 kb + (lane >> 4)], kb < K / 4 (0 for k >= 2L)
  f32x4 b0;     // b0[4 q + r]
  f32x4 w1;     // w1[4 q + r]
  float b1;
};
template <int L>
__device__ __forceinline__ PropMlp prop_mlp_load(const PropNet& n, float* encT, int lane) {
  constexpr int K = 2 * L, KB = (K + 3) / 4;
  const int m = lane & 15, q = lane >> 4;
  PropMlp w;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) w.a[kb] = (kb < KB && 4 * kb + q < K) ? n.w0[m * K + 4 * kb + q] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w.b0[r] = n.b0[4 * q + r];
    w.w1[r] = n.w1[4 * q + r];
  }
  w.b1 = n.b1[0];
  // the padding rows of the encoding block are read as B operands (against zero weights): keep them finite
#pragma unroll
  for (int k = K; k < 4 * KB; ++k) encT[k * PROP_ENC_STRIDE + lane] = 0.f;
  return w;
}
__device__ __forceinline__ float xor_lane(float v, int mask) { return __shfl_xor(v, mask, 64); }
template <int L>
__device__ __forceinline__ float prop_mlp_mfma(const PropMlp& w, float* encT, const float (&enc)[2 * L], int lane) {
  constexpr int K = 2 * L, KB = (K + 3) / 4;
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int k = 0; k < K; ++k) encT[k * PROP_ENC_STRIDE + lane] = enc[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float part[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 acc = w.b0;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a[kb], encT[(4 * kb + q) * PROP_ENC_STRIDE + 16 * t + i], acc, 0, 0, 0);
    float sdot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) sdot = fmaf(w.w1[r], fmaxf(acc[r], 0.f), sdot);
    part[t] = sdot;  // hidden units 4q .. 4q+3 of evaluation 16 t + i
  }
  __builtin_amdgcn_wave_barrier();  // (the block is rewritten by the next call)
  // butterfly over the lane groups q: lane (i, q) ends with the sum over all 16 hidden units of evaluation 16 q + i
  const bool q0 = (q & 1) != 0, q1 = (q & 2) != 0;
  // step 1 (partner q ^ 1): keep the tiles t with (t & 1) == (q & 1)
  const float send_a = q0 ? part[0] : part[1], send_b = q0 ? part[2] : part[3];
  const float keep_a = q0 ? part[1] : part[0], keep_b = q0 ? part[3] : part[2];
  const float ra = keep_a + xor_lane(send_a, 16), rb = keep_b + xor_lane(send_b, 16);  // tiles (q & 1), 2 + (q & 1)
  // step 2 (partner q ^ 2): keep the tile with bit 1 == (q & 2)
  const float send = q1 ? ra : rb, keep = q1 ? rb : ra;
  return w.b1 + (keep + xor_lane(send, 32));
}

// fp16 matrix mode of the sampler (cn_proposal_sample_mp with CN_MATRIX_F16 on half tables): tiny-cuda-nn's arithmetic class for the
// proposal networks too -- the grid interpolated on packed fp16 pairs (hash_level_pk), the 2L -> 16 layer ONE
// v_mfma_f32_16x16x16_f16 per 16-column tile (fp16 weights and inputs, fp32 accumulation; k = 2 level + feature, so a lane group's
// four k are the packed words of two levels as they stand), the hidden units rounded to fp16, the output rounded to fp16 as a
// tcnn network returns it.  Half the blend instructions and an eighth of the matrix cycles of the fp32 form.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
struct PropMlpH {
  f16x4 a;     // W0[m = lane & 15][k = 4 q .. 4 q + 3] as fp16 (0 for k >= 2L)
  f32x4 b0;    // b0[4 q + r]
  f32x4 w1;    // w1[4 q + r], fp16 values
  float b1;
};
template <int L>
__device__ __forceinline__ PropMlpH prop_mlp_load_h(const PropNet& n, unsigned* encw, int lane) {
  constexpr int K = 2 * L;
  static_assert(K <= 16, "one K = 16 block");
  const int m = lane & 15, q = lane >> 4;
  PropMlpH w;
#pragma unroll
  for (int e = 0; e < 4; ++e) w.a[e] = (4 * q + e < K) ? (_Float16)n.w0[m * K + 4 * q + e] : (_Float16)0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w.b0[r] = n.b0[4 * q + r];
    w.w1[r] = (float)(_Float16)n.w1[4 * q + r];
  }
  w.b1 = n.b1[0];
#pragma unroll
  for (int l = L; l < 8; ++l) encw[l * PROP_ENC_STRIDE + lane] = 0u;  // the levels the K block has room for and the grid lacks
  return w;
}
template <int L>
__device__ __forceinline__ float prop_mlp_f16(const PropMlpH& w, unsigned* encw, const unsigned (&featp)[L], int lane) {
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int l = 0; l < L; ++l) encw[l * PROP_ENC_STRIDE + lane] = featp[l];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float part[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned lo = encw[(2 * q) * PROP_ENC_STRIDE + 16 * t + i], hi = encw[(2 * q + 1) * PROP_ENC_STRIDE + 16 * t + i];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 words = {lo, hi};
    const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16f16(w.a, __builtin_bit_cast(f16x4, words), w.b0, 0, 0, 0);
    float sdot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) sdot = fmaf(w.w1[r], (float)(_Float16)fmaxf(acc[r], 0.f), sdot);
    part[t] = sdot;
  }
  __builtin_amdgcn_wave_barrier();
  const bool q0 = (q & 1) != 0, q1 = (q & 2) != 0;
  const float send_a = q0 ? part[0] : part[1], send_b = q0 ? part[2] : part[3];
  const float keep_a = q0 ? part[1] : part[0], keep_b = q0 ? part[3] : part[2];
  const float ra = keep_a + xor_lane(send_a, 16), rb = keep_b + xor_lane(send_b, 16);
  const float send = q1 ? ra : rb, keep = q1 ? rb : ra;
  return (float)(_Float16)(w.b1 + (keep + xor_lane(send, 32)));
}
template <int L>
__device__ __forceinline__ float prop_density_f16(const PropNet& n, const PropMlpH& mlp, unsigned* encw, int lane, const SceneDev& sc,
                                                  float px, float py, float pz) {
  const bool sel = normalize_position(sc, px, py, pz);
  unsigned featp[L];
#pragma unroll
  for (int l = 0; l < L; ++l)  // pinned level by level: left alone hipcc sinks every blend behind the last level's loads and spills
    featp[l] = pk_pin(hash_level_pk<true>(n.grid.table, n.grid.level(l), n.grid.pos_offset, px, py, pz));
  return expf(prop_mlp_f16<L>(mlp, encw, featp, lane)) * (sel ? 1.f : 0.f);
}

#ifndef CN_PROP_MLP_MFMA
#define CN_PROP_MLP_MFMA 1
#endif
template <int L, int H, bool HALF>
__device__ __forceinline__ float prop_density(const PropNet& n, const PropMlp& mlp, float* encT, int lane, const SceneDev& sc, float px,
                                              float py, float pz) {
  bool sel = normalize_position(sc, px, py, pz);
#if CN_PROP_ABLATE == 1  // timing only: no network at all
  return (px + py + pz) * (sel ? 1.f : 0.f);
#endif
  float enc[2 * L];
  // (Software-pipelining the levels -- level l + 1's loads issued before level l is blended, hash_level_xpair_issue / _blend --
  //  measured slower: 0.85 vs 0.67 ms per C2 launch, the second level in flight costs 20 registers and spills.)
#pragma unroll
  for (int l = 0; l < L; ++l) {
#if CN_PROP_XPAIR
    float2 f = hash_level_xpair<HALF>(n.grid.table, n.grid.level(l), n.grid.pos_offset, px, py, pz);
#else
    float2 f = hash_level<HALF>(n.grid.table, n.grid.level(l), n.grid.pos_offset, px, py, pz);
#endif
    enc[2 * l] = f.x;
    enc[2 * l + 1] = f.y;
  }
#if CN_PROP_ABLATE == 2  // timing only: the encoding without the MLP
  {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) acc += enc[k];
    return expf(acc) * (sel ? 1.f : 0.f);
  }
#endif
#if CN_PROP_MLP_MFMA
  static_assert(H == 16, "one 16-row tile of hidden units");
  const float out = prop_mlp_mfma<L>(mlp, encT, enc, lane);
#else
  float out = n.b1[0];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float a = n.b0[h];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) a = fmaf(n.w0[h * 2 * L + k], enc[k], a);
    out = fmaf(n.w1[h], fmaxf(a, 0.f), out);
  }
#endif
  return expf(out) * (sel ? 1.f : 0.f);
}

template <bool HALF>
__device__ __forceinline__ float prop_density_dispatch(const PropNet& n, const PropMlp& mlp, float* encT, int lane, const SceneDev& sc,
                                                       float px, float py, float pz) {
  if (n.grid.num_levels == 5) return prop_density<5, 16, HALF>(n, mlp, encT, lane, sc, px, py, pz);
  return prop_density<7, 16, HALF>(n, mlp, encT, lane, sc, px, py, pz);
}

// HALF: the proposal nets' hash tables hold half2 entries (CN_TABLE_F16).  TRAIN: the training forward of
// ProposalNetworkSampler (fruit_nerf.py:549 under model.train()): stratified single-jitter level-0 bins
// (ray_samplers.py:84-87), PDF resampling at u + rand / nb, and every level's bins / intervals / densities written out.
#ifndef CN_PROP_SAMPLE_WAVES
#define CN_PROP_SAMPLE_WAVES 4  // waves per SIMD the register allocation is held to (the kernel is latency-bound: one wave per ray)
#endif
template <bool HALF, bool TRAIN, bool F16 = false>
__global__ void __launch_bounds__(256, CN_PROP_SAMPLE_WAVES) proposal_sample_kernel(PropArgs A) {
  static_assert(!F16 || (HALF && !TRAIN), "the fp16 mode evaluates half tables, eval only");
  extern __shared__ __align__(16) float lds[];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int stride = 4 * A.smax + 4;
  float* bins_a = lds + wave * stride;    // [smax+1] spacing bins of the current level
  float* bins_b = bins_a + A.smax + 1;    // [smax+1] next level
  float* wts = bins_b + A.smax + 1;       // [smax]
  float* cdf = wts + A.smax;              // [smax+1]
  float* encT = lds + 4 * stride + wave * (A.enc_rows * PROP_ENC_STRIDE);  // this wave's encoding block of prop_mlp_mfma
  const long long waves = (long long)gridDim.x * 4;
  for (long long rr = blockIdx.x * 4LL + wave; rr < A.num_rays; rr += waves) {
    const long long r = __builtin_amdgcn_readfirstlane((int)rr);
    const float ox = A.origins[3 * r], oy = A.origins[3 * r + 1], oz = A.origins[3 * r + 2];
    const float dx = A.directions[3 * r], dy = A.directions[3 * r + 1], dz = A.directions[3 * r + 2];
    const float sn = spacing_fn(CN_SPACING_PIECEWISE, A.nears[r]), sf = spacing_fn(CN_SPACING_PIECEWISE, A.fars[r]);
    float* cur = bins_a;
    float* nxt = bins_b;
    // level 0 bins: linspace(0,1,S0+1) in the spacing domain
    {
      const int s0 = A.s_prop[0];
      if constexpr (TRAIN) {
        const float t = A.jitter[r];
        for (int e = lane; e <= s0; e += 64) {
          const float be = linspace01(e, s0 + 1);
          const float lo = e == 0 ? be : (be + linspace01(e - 1, s0 + 1)) / 2.f;
          const float hi = e == s0 ? be : (linspace01(e + 1, s0 + 1) + be) / 2.f;
          cur[e] = lo + (hi - lo) * t;
        }
      } else {
        for (int e = lane; e <= s0; e += 64) cur[e] = linspace01(e, s0 + 1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int lvl = 0; lvl < A.num_levels; ++lvl) {
      const int S = A.s_prop[lvl];
      const PropNet& net = A.net[lvl];
      PropMlp mlp{};
      PropMlpH mlph{};
      if constexpr (F16) {
        mlph = net.grid.num_levels == 5 ? prop_mlp_load_h<5>(net, reinterpret_cast<unsigned*>(encT), lane)
                                        : prop_mlp_load_h<7>(net, reinterpret_cast<unsigned*>(encT), lane);
      } else {
        mlp = net.grid.num_levels == 5 ? prop_mlp_load<5>(net, encT, lane) : prop_mlp_load<7>(net, encT, lane);
      }
      CompositeState st;
      for (int c0 = 0; c0 < S; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < S;
        const int ic = valid ? i : S - 1;
        const float t0 = spacing_to_euclid(CN_SPACING_PIECEWISE, cur[ic], sn, sf);
        const float t1 = spacing_to_euclid(CN_SPACING_PIECEWISE, cur[ic + 1], sn, sf);
        const float mid = (t0 + t1) / 2.f;
        float den;
        if constexpr (F16) {
          unsigned* encw = reinterpret_cast<unsigned*>(encT);
          den = net.grid.num_levels == 5 ? prop_density_f16<5>(net, mlph, encw, lane, A.scene, ox + dx * mid, oy + dy * mid, oz + dz * mid)
                                         : prop_density_f16<7>(net, mlph, encw, lane, A.scene, ox + dx * mid, oy + dy * mid, oz + dz * mid);
        } else {
          den = prop_density_dispatch<HALF>(net, mlp, encT, lane, A.scene, ox + dx * mid, oy + dy * mid, oz + dz * mid);
        }
        const float w = composite_chunk(st, valid, i == S - 1, t1 - t0, den, mid, 0.f, 0.f, 0.f, 0.f, false);
        if (valid) wts[i] = w;
        if constexpr (TRAIN) {
          if (valid) {
            const long long o = r * S + i;
            A.lv_starts[lvl][o] = t0;
            A.lv_ends[lvl][o] = t1;
            A.lv_density[lvl][o] = den;
          }
        }
      }
      if constexpr (TRAIN) {
        for (int e = lane; e <= S; e += 64) A.lv_sp[lvl][r * (S + 1) + e] = cur[e];
      }
      if (A.out_depth && lane == 0) A.out_depth[lvl * A.num_rays + r] = st.found ? st.depth : st.last_mid;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      wave_cdf_from_weights(wts, S, A.anneal_dev ? *A.anneal_dev : A.anneal, cdf);
      const int s_next = lvl + 1 < A.num_levels ? A.s_prop[lvl + 1] : A.s_final;
      const int nb = s_next + 1;
      const bool last = lvl + 1 == A.num_levels;
      for (int b = lane; b < nb; b += 64) {
        float bin = pdf_invert(cdf, cur, S, pdf_u(b, nb, TRAIN ? A.jitter + (lvl + 1) * A.num_rays + r : nullptr, 1));
        nxt[b] = bin;
        if (last) {
          if (A.out_sp) A.out_sp[r * nb + b] = bin;
          const float eu = spacing_to_euclid(CN_SPACING_PIECEWISE, bin, sn, sf);
          A.out_eu[r * nb + b] = eu;
          if constexpr (TRAIN) {
            if (A.final_starts && b < s_next) A.final_starts[r * s_next + b] = eu;
            if (A.final_ends && b > 0) A.final_ends[r * s_next + b - 1] = eu;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float* t = cur;
      cur = nxt;
      nxt = t;
    }
  }
}

int validate_grid(const cn_grid& g, const char* name);  // field_simple.hip

}  // namespace cn

extern "C" size_t cn_proposal_sample_workspace_bytes(int64_t, const int32_t*, int32_t, int32_t) { return 0; }

namespace cn {
static int proposal_sample_launch(const char* who, const cn_density_params* const* props, int32_t num_levels,
                                  const cn_scene* scene, const float* origins, const float* directions,
                                  const float* nears, const float* fars, int64_t num_rays, const int32_t* s_prop,
                                  int32_t s_final, float anneal, const float* jitter,
                                  const cn_proposal_level_out* levels, float* euclidean_bins, float* spacing_bins,
                                  float* prop_depth, float* final_starts, float* final_ends, cn_stream_t stream,
                                  int matrix_precision = CN_MATRIX_FP32, const float* anneal_dev = nullptr) {
  CN_REQUIRE(props && scene && origins && directions && nears && fars && s_prop && euclidean_bins, CN_ERR_INVALID,
             "%s: null argument", who);
  CN_REQUIRE(matrix_precision == CN_MATRIX_FP32 || matrix_precision == CN_MATRIX_SPLIT_BF16 || matrix_precision == CN_MATRIX_F16,
             CN_ERR_INVALID, "%s: matrix_precision %d", who, matrix_precision);
  CN_REQUIRE(num_levels >= 1 && num_levels <= PROP_MAX_LEVELS, CN_ERR_UNSUPPORTED,
             "%s: %d proposal iterations (max %d)", who, num_levels, PROP_MAX_LEVELS);
  CN_REQUIRE(s_final >= 1 && s_final <= PROP_MAX_SAMPLES, CN_ERR_UNSUPPORTED, "%s: s_final %d", who, s_final);
  CN_REQUIRE(num_rays < (1LL << 31), CN_ERR_INVALID, "%s: at most 2^31-1 rays per call", who);
  const bool train = jitter != nullptr;
  CN_REQUIRE(!train || levels, CN_ERR_INVALID, "%s: null level outputs", who);
  PropArgs A{};
  A.smax = s_final;
  for (int l = 0; l < num_levels; ++l) {
    CN_REQUIRE(props[l], CN_ERR_INVALID, "%s: null proposal net %d", who, l);
    CN_REQUIRE(s_prop[l] >= 1 && s_prop[l] <= PROP_MAX_SAMPLES, CN_ERR_UNSUPPORTED,
               "%s: %d samples at level %d (max %d)", who, s_prop[l], l, PROP_MAX_SAMPLES);
    const cn_density_params& p = *props[l];
    int rc = validate_grid(p.grid, "proposal grid");
    if (rc) return rc;
    bool ok = (p.grid.num_levels == 5 || p.grid.num_levels == 7) && p.mlp.num_layers == 2 &&
              p.mlp.dims[0] == 2 * p.grid.num_levels && p.mlp.dims[1] == 16 && p.mlp.dims[2] == 1;
    CN_REQUIRE(ok, CN_ERR_UNSUPPORTED, "%s: proposal net %d is not {5|7 levels, 2L->16->1}; compose the unfused calls",
               who, l);
    CN_REQUIRE(p.mlp.weight[0] && p.mlp.bias[0] && p.mlp.weight[1] && p.mlp.bias[1], CN_ERR_INVALID,
               "%s: null MLP parameter in net %d", who, l);
    A.net[l].grid = make_grid_dev(p.grid);
    CN_REQUIRE(A.net[l].grid.half == A.net[0].grid.half, CN_ERR_UNSUPPORTED,
               "%s: the proposal nets' hash tables must share one dtype", who);
    A.net[l].w0 = p.mlp.weight[0];
    A.net[l].b0 = p.mlp.bias[0];
    A.net[l].w1 = p.mlp.weight[1];
    A.net[l].b1 = p.mlp.bias[1];
    A.s_prop[l] = s_prop[l];
    if (s_prop[l] > A.smax) A.smax = s_prop[l];
    if (train) {
      const cn_proposal_level_out& o = levels[l];
      CN_REQUIRE(o.spacing_bins && o.starts && o.ends && o.density, CN_ERR_INVALID, "%s: null output of level %d", who, l);
      A.lv_sp[l] = o.spacing_bins;
      A.lv_starts[l] = o.starts;
      A.lv_ends[l] = o.ends;
      A.lv_density[l] = o.density;
    }
  }
  if (num_rays <= 0) return CN_OK;
  A.num_levels = num_levels;
  A.s_final = s_final;
  A.anneal = anneal;
  A.anneal_dev = anneal_dev;
  A.scene = make_scene_dev(*scene);
  A.origins = origins;
  A.directions = directions;
  A.nears = nears;
  A.fars = fars;
  A.num_rays = num_rays;
  A.out_eu = euclidean_bins;
  A.out_sp = spacing_bins;
  A.out_depth = prop_depth;
  A.jitter = jitter;
  A.final_starts = final_starts;
  A.final_ends = final_ends;
  A.enc_rows = 0;
  for (int l = 0; l < num_levels; ++l) A.enc_rows = std::max(A.enc_rows, (2 * A.net[l].grid.num_levels + 3) / 4 * 4);
  const size_t lds = ((size_t)4 * (4 * A.smax + 4) + (size_t)4 * A.enc_rows * PROP_ENC_STRIDE) * sizeof(float);
  const dim3 grid(grid_for(num_rays, 4, 256 * 8)), block(256);
  const bool half = A.net[0].grid.half;
#define CN_PROP_LAUNCH(H, T) hipLaunchKernelGGL((proposal_sample_kernel<H, T>), grid, block, lds, as_stream(stream), A)
  // fp16 mode: half tables only (a float table keeps the fp32 form: the mode is the arithmetic of an imported tcnn model; the
  // split-bf16 option has nothing to split here -- the sampler's matrix work is a tenth of its instructions -- and runs fp32 too)
  if (!train && half && matrix_precision == CN_MATRIX_F16) {
    hipLaunchKernelGGL((proposal_sample_kernel<true, false, true>), grid, block, lds, as_stream(stream), A);
  } else
  if (train) {
    if (half) CN_PROP_LAUNCH(true, true); else CN_PROP_LAUNCH(false, true);
  } else {
    if (half) CN_PROP_LAUNCH(true, false); else CN_PROP_LAUNCH(false, false);
  }
#undef CN_PROP_LAUNCH
  return check_launch(who);
}
}  // namespace cn

extern "C" int cn_proposal_sample(const cn_density_params* const* props, int32_t num_levels, const cn_scene* scene,
                                  const float* origins, const float* directions, const float* nears, const float* fars,
                                  int64_t num_rays, const int32_t* s_prop, int32_t s_final, float anneal,
                                  float* euclidean_bins, float* spacing_bins, float* prop_depth, void*, size_t,
                                  cn_stream_t stream) {
  return cn::proposal_sample_launch("cn_proposal_sample", props, num_levels, scene, origins, directions, nears, fars,
                                    num_rays, s_prop, s_final, anneal, nullptr, nullptr, euclidean_bins, spacing_bins,
                                    prop_depth, nullptr, nullptr, stream);
}

extern "C" int cn_proposal_sample_mp(const cn_density_params* const* props, int32_t num_levels, const cn_scene* scene,
                                     const float* origins, const float* directions, const float* nears, const float* fars,
                                     int64_t num_rays, const int32_t* s_prop, int32_t s_final, float anneal,
                                     float* euclidean_bins, float* spacing_bins, float* prop_depth, int32_t matrix_precision,
                                     cn_stream_t stream) {
  return cn::proposal_sample_launch("cn_proposal_sample_mp", props, num_levels, scene, origins, directions, nears, fars,
                                    num_rays, s_prop, s_final, anneal, nullptr, nullptr, euclidean_bins, spacing_bins,
                                    prop_depth, nullptr, nullptr, stream, matrix_precision);
}

extern "C" int cn_proposal_sample_train(const cn_density_params* const* props, int32_t num_levels,
                                        const cn_scene* scene, const float* origins, const float* directions,
                                        const float* nears, const float* fars, int64_t num_rays, const int32_t* s_prop,
                                        int32_t s_final, float anneal, const float* jitter,
                                        const cn_proposal_level_out* levels, float* euclidean_bins, float* spacing_bins,
                                        float* final_starts, float* final_ends, cn_stream_t stream) {
  CN_REQUIRE(jitter, CN_ERR_INVALID, "cn_proposal_sample_train: null jitter");
  return cn::proposal_sample_launch("cn_proposal_sample_train", props, num_levels, scene, origins, directions, nears,
                                    fars, num_rays, s_prop, s_final, anneal, jitter, levels, euclidean_bins,
                                    spacing_bins, nullptr, final_starts, final_ends, stream);
}

extern "C" int cn_proposal_sample_train_dev(const cn_density_params* const* props, int32_t num_levels,
                                            const cn_scene* scene, const float* origins, const float* directions,
                                            const float* nears, const float* fars, int64_t num_rays, const int32_t* s_prop,
                                            int32_t s_final, const float* anneal_dev, const float* jitter,
                                            const cn_proposal_level_out* levels, float* euclidean_bins, float* spacing_bins,
                                            float* final_starts, float* final_ends, cn_stream_t stream) {
  CN_REQUIRE(jitter && anneal_dev, CN_ERR_INVALID, "cn_proposal_sample_train_dev: null jitter / anneal");
  return cn::proposal_sample_launch("cn_proposal_sample_train_dev", props, num_levels, scene, origins, directions, nears,
                                    fars, num_rays, s_prop, s_final, 1.f, jitter, levels, euclidean_bins, spacing_bins,
                                    nullptr, final_starts, final_ends, stream, CN_MATRIX_FP32, anneal_dev);
}

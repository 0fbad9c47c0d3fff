// FruitField backward on the fp32 matrix cores (included by train_field.hip, after FieldBwdArgs and the hash helpers).
//
// One 512-thread workgroup (8 waves, 2 per SIMD) per CU walks 32-sample tiles.  Per tile the forward is recomputed and
// every activation / delta lives in LDS as [feature][36] (32 samples + 4 pad: row stride = 4 banks, so the 16-byte and
// 4-byte operand reads below are conflict-free); the MLP weights sit in LDS for the whole kernel as [out][in + 4].
// Three v_mfma_f32_16x16x4_f32 block routines do all the dense work (exact fp32 FMA chains, like the forward kernel):
//   blk_fwd  Y[n][s]  = sum_k W[n][k] X[k][s]      A = W rows (16-byte reads along k),     B = X rows
//   blk_bwd  dX[k][s] = sum_n W[n][k] dY[n][s]     A = W columns,                          B = dY rows
//   blk_dw   dW[n][k] += sum_s dY[n][s] X[k][s]    A = dY rows, B = X rows (both 16-byte reads along s)
// A wave owns one 16x16 output block per phase ((row tile, column tile) = (wave >> 1, wave & 1)) and 9 fixed 16x16
// blocks of the weight gradients, whose accumulators stay in registers across all tiles of the workgroup and are
// flushed with one atomic per entry at the end.  The semantic and the colour branch run one after the other through the
// same four LDS buffers (548 rows x 144 B + 81 KB of weights = 159.9 KB of the 160 KB LDS).
//
// Phases (a workgroup barrier between each):
//   gather | h1 | o16 | s1 | s2 | d_s2 (+dW head) | d_s1 (+dW sem1) | c1 (+dW sem0) | c2 | rgb | d_c2 (+dW rgb) |
//   d_c1 (+dW col1) | d_cin (+dW col0) | d_o16, embedding / SH gradients | d_h1 | d_enc (+dW base1) |
//   hash scatter (+dW base0) | position gradient
#pragma once

namespace cn {
namespace mf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TSM = 32;   // samples per tile
constexpr int LDA = 36;   // activation row stride (floats)
constexpr int NT = 512;   // threads per workgroup

// weight image (floats): [out][in + 4]
constexpr int W_B0 = 0;                    // 64 x 36   base 32 -> 64
constexpr int W_B1 = W_B0 + 64 * 36;       // 16 x 68   base 64 -> 16
constexpr int W_S0 = W_B1 + 16 * 68;       // 64 x 20   semantic 15(+1 zero) -> 64
constexpr int W_S1 = W_S0 + 64 * 20;       // 64 x 68   semantic 64 -> 64
constexpr int W_C0 = W_S1 + 64 * 68;       // 64 x 68   colour 63(+1 zero) -> 64
constexpr int W_C1 = W_C0 + 64 * 68;       // 64 x 68   colour 64 -> 64
constexpr int W_RGB = W_C1 + 64 * 68;      // 16 x 68   rows 0..2 = colour head, rest zero
constexpr int W_SEM = W_RGB + 16 * 68;     // 16 x 68   row 0 = semantic head, rest zero
constexpr int B_0 = W_SEM + 16 * 68;       // biases
constexpr int B_1 = B_0 + 64;
constexpr int B_S0 = B_1 + 16;
constexpr int B_S1 = B_S0 + 64;
constexpr int B_C0 = B_S1 + 64;
constexpr int B_C1 = B_C0 + 64;
constexpr int B_RGB = B_C1 + 64;           // 16 (3 real)
constexpr int T_SCALE = B_RGB + 16;        // 16 level scales (a per-lane index into the kernarg array would go to scratch)
constexpr int W_END = T_SCALE + 16;
// activation rows
constexpr int A_ENC = 0;                   // 32
constexpr int A_H1 = A_ENC + 32;           // 64
constexpr int A_O16 = A_H1 + 64;           // 20: rows 0..15 = logit | geo, rows 16..19 stay zero (pad of the 15-wide input)
constexpr int A_CIN = A_O16 + 20;          // 64: SH 0..15 | geo 16..30 | appearance 31..62 | row 63 stays zero
constexpr int A_A1 = A_CIN + 64;           // 64: s1, later c1
constexpr int A_A2 = A_A1 + 64;            // 64: s2, later c2
constexpr int A_D2 = A_A2 + 64;            // 64: d_s2, d_c2, d_h1
constexpr int A_D1 = A_D2 + 64;            // 64: d_s1, d_c1, d_enc
constexpr int A_DCIN = A_D1 + 64;          // 64
constexpr int A_DO16 = A_DCIN + 64;        // 16
constexpr int A_DRGB = A_DO16 + 16;        // 16: rows 0..2 written, rest stay zero
constexpr int A_DSEM = A_DRGB + 16;        // 16: row 0 written, rest stay zero
constexpr int A_ROWS = A_DSEM + 16;        // 548
constexpr size_t LDS_BYTES = (size_t)(W_END + A_ROWS * LDA) * sizeof(float);

// The lane coordinates go through an empty volatile asm at every use site: otherwise LICM hoists the (loop-invariant)
// LDS addresses of all ~40 block calls out of the tile loop and keeps them live across it -- 126 spilled VGPRs.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
#define CN_LANE_IQ const int lane_o = opaque(lane); const int i = lane_o & 15, q = lane_o >> 4;

// Matrix modes of the kernel (template parameter MM): 0 = exact fp32 (v_mfma_f32_16x16x4_f32, the chains above), 1 = the
// reference's mixed-precision class (cn_field_backward_mp, matrix_precision = CN_MATRIX_F16): the forward recompute on
// v_mfma_f32_16x16x16_f16 -- weights and layer inputs rounded to fp16 as they leave LDS, fp32 accumulation, which is what the
// fp16 render mode computes (render_f16.hpp) and tiny-cuda-nn's FullyFusedMLP under mixed_precision=True -- and the two
// gradient products (dX, dW) on v_mfma_f32_16x16x16_bf16: deltas are ~1e-9 at 65 536 rays, below fp16's normal range, and
// bf16 keeps fp32's exponent, so no loss scale runs through the kernel (tcnn scales its fp16 gradients by 128, nerfstudio's
// GradScaler by 2^10 and up).  The LDS images stay fp32: a lane's four k-steps of the fp32 chain ARE the four-element operand
// of the 16 x 16 x 16 instruction, so the reads are the same and one matrix instruction replaces four.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_f16x4(const f32x4& a, const f32x4& b, f32x4 acc) {
  f16x4 ah, bh;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ah[e] = (_Float16)a[e];
    bh[e] = (_Float16)b[e];
  }
  return __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16x4(const f32x4& a, const f32x4& b, f32x4 acc) {
  bf16x4 ah, bh;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ah[e] = (__bf16)a[e];
    bh[e] = (__bf16)b[e];
  }
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bh), acc, 0, 0, 0);
}

// (Two independent accumulation chains per block -- even / odd k blocks, summed at the end -- measured slower: 17.86 vs 17.61 ms
//  per iteration at 65 536 rays; the second wave of the SIMD already fills the gaps of a dependent chain.)
// CN_ABL_QUARTER_MFMA (timing only, wrong results): the operand reads of four k-steps and ONE fp32 matrix instruction -- the
// probe that priced the 16-bit mode before it was built.
template <int K, int MM>
__device__ __forceinline__ f32x4 blk_fwd(const float* W, int ws, int n0, const float* X, int s0, f32x4 acc, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int kb = 0; kb < K / 16; ++kb) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(W + (n0 + i) * ws + 16 * kb + 4 * q);
    if (MM == 1) {
      f32x4 b;
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = X[(16 * kb + 4 * q + e) * LDA + s0 + i];
      acc = mfma_f16x4(a, b, acc);
      continue;
    }
#ifdef CN_ABL_QUARTER_MFMA
    float bsum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum += X[(16 * kb + 4 * q + e) * LDA + s0 + i];
    acc = CN_MFMA(a[0] + a[1] + a[2] + a[3], bsum, acc);
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float av = a[e];
      const float b = X[(16 * kb + 4 * q + e) * LDA + s0 + i];
      acc = CN_MFMA(av, b, acc);
    }
#endif
  }
  return acc;
}

template <int N, int MM>
__device__ __forceinline__ f32x4 blk_bwd(const float* W, int ws, int k0, const float* dY, int s0, f32x4 acc, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int nb = 0; nb < N / 16; ++nb) {
    if (MM == 1) {
      f32x4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 16 * nb + 4 * q + e;
        a[e] = W[n * ws + k0 + i];
        b[e] = dY[n * LDA + s0 + i];
      }
      acc = mfma_bf16x4(a, b, acc);
      continue;
    }
#ifdef CN_ABL_QUARTER_MFMA
    float asum = 0.f, bsum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 16 * nb + 4 * q + e;
      asum += W[n * ws + k0 + i];
      bsum += dY[n * LDA + s0 + i];
    }
    acc = CN_MFMA(asum, bsum, acc);
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 16 * nb + 4 * q + e;
      const float a = W[n * ws + k0 + i];
      const float b = dY[n * LDA + s0 + i];
      acc = CN_MFMA(a, b, acc);
    }
#endif
  }
  return acc;
}

template <int MM>
__device__ __forceinline__ f32x4 blk_dw(const float* dY, int n0, const float* X, int k0, f32x4 acc, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int sb = 0; sb < TSM / 16; ++sb) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(dY + (n0 + i) * LDA + 16 * sb + 4 * q);
    const f32x4 b = *reinterpret_cast<const f32x4*>(X + (k0 + i) * LDA + 16 * sb + 4 * q);
    if (MM == 1) {
      acc = mfma_bf16x4(a, b, acc);
      continue;
    }
#ifdef CN_ABL_QUARTER_MFMA
    acc = CN_MFMA(a[0] + a[1] + a[2] + a[3], b[0] + b[1] + b[2] + b[3], acc);
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float av = a[e], bv = b[e];
      acc = CN_MFMA(av, bv, acc);
    }
#endif
  }
  return acc;
}

__device__ __forceinline__ f32x4 bias4(const float* b, int n0, int lane) {
  const int q = opaque(lane) >> 4;
  f32x4 v;
  v[0] = b[n0 + 4 * q + 0];
  v[1] = b[n0 + 4 * q + 1];
  v[2] = b[n0 + 4 * q + 2];
  v[3] = b[n0 + 4 * q + 3];
  return v;
}

// lane (q, i) holds rows n0 + 4q + r of column s0 + i
template <bool RELU>
__device__ __forceinline__ void store_blk(float* out, int n0, int s0, f32x4 v, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = v[r];
    if (RELU) x = fmaxf(x, 0.f);
    out[(n0 + 4 * q + r) * LDA + s0 + i] = x;
  }
}

// delta block: optional ReLU gate by the forward activation, accumulate the bias gradient partials, store
__device__ __forceinline__ void store_delta(float* out, const float* act, int n0, int s0, f32x4 v, f32x4& gb, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = v[r];
    if (act) x = act[(n0 + 4 * q + r) * LDA + s0 + i] > 0.f ? x : 0.f;
    gb[r] += x;
    out[(n0 + 4 * q + r) * LDA + s0 + i] = x;
  }
}

__device__ __forceinline__ void flush_dw(float* g, int N, int K, int n0, int k0, f32x4 acc, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + 4 * q + r, k = k0 + i;
    if (n < N && k < K && acc[r] != 0.f) cn_atomic_add(g + n * K + k, acc[r]);
  }
}

// sum over the 16 lanes of a DPP row, valid in lane i == 15 of each row... use a plain xor-free row reduction
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0x111>(v);  // row_shr:1
  v += dpp_f32<0x112>(v);  // row_shr:2
  v += dpp_f32<0x114>(v);  // row_shr:4
  v += dpp_f32<0x118>(v);  // row_shr:8
  return v;                // lane 15 of the row holds the total
}

__device__ __forceinline__ void flush_bias(float* g, int n0, f32x4 gb, int lane) {
  CN_LANE_IQ
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s = row16_sum(gb[r]);
    if (i == 15 && s != 0.f) cn_atomic_add(g + n0 + 4 * q + r, s);
  }
}

template <int MM>
__global__ void __launch_bounds__(NT) field_backward_mfma_kernel(FieldBwdArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* act = lds + W_END;
  float* ENC = act + A_ENC * LDA;
  float* H1 = act + A_H1 * LDA;
  float* O16 = act + A_O16 * LDA;
  float* CIN = act + A_CIN * LDA;
  float* A1 = act + A_A1 * LDA;
  float* A2 = act + A_A2 * LDA;
  float* D2 = act + A_D2 * LDA;
  float* D1 = act + A_D1 * LDA;
  float* DCIN = act + A_DCIN * LDA;
  float* DO16 = act + A_DO16 * LDA;
  float* DRGB = act + A_DRGB * LDA;
  float* DSEM = act + A_DSEM * LDA;

  // ---- weight image + constant zero rows ----------------------------------------------------------------------------
  for (int e = tid; e < W_END + A_ROWS * LDA; e += NT) lds[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < 64 * 32; e += NT) lds[W_B0 + (e >> 5) * 36 + (e & 31)] = A.p.w0[e];
  for (int e = tid; e < 16 * 64; e += NT) lds[W_B1 + (e >> 6) * 68 + (e & 63)] = A.p.w1[e];
  for (int e = tid; e < 64 * 15; e += NT) lds[W_S0 + (e / 15) * 20 + (e % 15)] = A.p.ws0[e];
  for (int e = tid; e < 64 * 64; e += NT) lds[W_S1 + (e >> 6) * 68 + (e & 63)] = A.p.ws1[e];
  for (int e = tid; e < 64 * 63; e += NT) lds[W_C0 + (e / 63) * 68 + (e % 63)] = A.p.wc0[e];
  for (int e = tid; e < 64 * 64; e += NT) lds[W_C1 + (e >> 6) * 68 + (e & 63)] = A.p.wc1[e];
  for (int e = tid; e < 3 * 64; e += NT) lds[W_RGB + (e >> 6) * 68 + (e & 63)] = A.p.wc2[e];
  for (int e = tid; e < 64; e += NT) {
    lds[W_SEM + e] = A.p.wh[e];
    lds[B_0 + e] = A.p.b0[e];
    lds[B_S0 + e] = A.p.bs0[e];
    lds[B_S1 + e] = A.p.bs1[e];
    lds[B_C0 + e] = A.p.bc0[e];
    lds[B_C1 + e] = A.p.bc1[e];
    if (e < 16) lds[B_1 + e] = A.p.b1[e];
    if (e < 16) lds[T_SCALE + e] = A.grid.scale[e];
    if (e < 3) lds[B_RGB + e] = A.p.bc2[e];
  }
  __syncthreads();

  const float* Wb0 = lds + W_B0;
  const float* Wb1 = lds + W_B1;
  const float* Ws0 = lds + W_S0;
  const float* Ws1 = lds + W_S1;
  const float* Wc0 = lds + W_C0;
  const float* Wc1 = lds + W_C1;
  const float* Wrgb = lds + W_RGB;
  const float* Wsem = lds + W_SEM;

  const int rt = wave >> 1, ct = wave & 1;  // this wave's 16x16 output block of a 64-row layer
  const int n0 = 16 * rt, s0 = 16 * ct;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // weight-gradient accumulators (fixed ownership, see the flush at the end)
  f32x4 gS1[2] = {zero4, zero4}, gC1[2] = {zero4, zero4}, gC0[2] = {zero4, zero4}, gB0 = zero4, gX = zero4, gY = zero4;
  // bias-gradient partials of the delta blocks this wave produces
  f32x4 bS1 = zero4, bS0 = zero4, bC1 = zero4, bC0 = zero4, bH1 = zero4;
  float b_o16 = 0.f, b_rgb[3] = {0.f, 0.f, 0.f}, b_sem = 0.f;

  const int s = tid & 31, lvl = tid >> 5;  // gather / scatter role: one (sample, level) per thread
  const Lvl my_lv = lane_level(A.grid, lvl);
  const long long total = A.R * (long long)A.S;
  const long long ntiles = (total + TSM - 1) / TSM;
  // The hash scatter of a tile is DEFERRED to just after the next tile's gather: a wave's loads and atomics share one
  // in-order counter (vmcnt), so a gather issued after the atomics waits until every one of them has been acknowledged by
  // the memory side -- at the end of a tile that round trip was exposed on every CU; issued after the next gather, it
  // drains under that tile's matrix phases (LDS and MFMA only: the per-tile inputs are all loaded in the gather phase).
  bool have_prev = false;   // workgroup-uniform
  float p_px = 0.f, p_py = 0.f, p_pz = 0.f, p_wx = 0.f, p_wy = 0.f, p_wz = 0.f, p_self = 0.f, p_g0 = 0.f, p_g1 = 0.f;
  long long p_ismp = 0;
  bool p_valid = false;
  // (the loop makes one trip more than the workgroup has tiles: that trip only issues the last tile's scatter -- written
  //  once in the loop rather than as a lambda called twice, which hipcc keeps as a closure in scratch memory)
  // Tile -> workgroup: one contiguous run of tiles per workgroup (round 5; before: tile = blockIdx.x + k gridDim.x).  The tiles
  // in flight at one moment are then far apart in the batch, a workgroup's own consecutive tiles are neighbours.  For a batch in
  // draw order nothing changes (15.58 against 15.59 ms per 65 536-ray iteration); for a batch sorted by camera and pixel
  // (FruitDataManager.sort_batches) the interleaved form had every workgroup adding to its neighbours' table lines at the same
  // moment (22.99 ms), this one keeps the gathers' locality without the atomics' contention (14.94 ms).
  const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
  const long long t_end = ((long long)blockIdx.x + 1) * per < ntiles ? ((long long)blockIdx.x + 1) * per : ntiles;
  for (long long tile = blockIdx.x * per;; ++tile) {
    const bool live = tile < t_end;  // workgroup-uniform
    if (!live && !have_prev) break;
    // ---- gather ------------------------------------------------------------------------------------------------------
    const long long ismp = tile * TSM + s;
    const bool valid = live && ismp < total;
    const long long ic = valid ? ismp : total - 1;
    const long long r = ic / A.S;
    const float mid = (A.starts[ic] + A.ends[ic]) / 2.f;
    const float dirx = A.directions[3 * r], diry = A.directions[3 * r + 1], dirz = A.directions[3 * r + 2];
    const float wx = A.origins[3 * r] + dirx * mid, wy = A.origins[3 * r + 1] + diry * mid,
                wz = A.origins[3 * r + 2] + dirz * mid;
    float px = wx, py = wy, pz = wz;
    const float self = normalize_position(A.scene, px, py, pz) ? 1.f : 0.f;
    // (the Jacobian of the level's two features with respect to the normalised position comes out of the same eight corners:
    //  the position gradient of the tile is g0 * J.x + g1 * J.y at its end, no second gather -- hash_level_jac)
    v2f_t jx = {0.f, 0.f}, jy = {0.f, 0.f}, jz = {0.f, 0.f};
    if (live) {
      // (bit 64, timing only: no table gathers in the forward recompute)
      const float2 f = (A.debug_skip & 64) ? make_float2(px * 0.01f, py * 0.01f)
                                           : hash_level_jac(A.p.table, my_lv, A.grid.pos_offset, px, py, pz, jx, jy, jz);
      ENC[(2 * lvl) * LDA + s] = f.x;
      ENC[(2 * lvl + 1) * LDA + s] = f.y;
    }
    // every other global input of the tile is read here too, so that nothing between this phase and the next gather waits
    // on vmcnt (see deferred_scatter): the upstream colour gradients (two waves, 16 lanes each), the density gradient and
    // the camera row of the sample
    const long long cam_row = A.app_per_camera ? A.cam_idx[r] : 0;
    const float dd_in = (lvl == 0 && valid) ? A.d_density[ic] : 0.f;
    float drgb_in[3] = {0.f, 0.f, 0.f};
    if (wave < 2 && (lane >> 4) == 0) {
      const long long io = tile * TSM + 16 * wave + (lane & 15);
      if (io < total) {
        drgb_in[0] = A.d_rgb[3 * io];
        drgb_in[1] = A.d_rgb[3 * io + 1];
        drgb_in[2] = A.d_rgb[3 * io + 2];
      }
    }
    if (lvl == 0) {
      const float ds = valid ? A.d_sem[ic] : 0.f;
      DSEM[s] = ds;
      b_sem += ds;
    } else if (lvl == 1) {
      float dx = dirx, dy = diry, dz = dirz;
      if (!A.sh_unit) {
        dx = (dx + 1.f) / 2.f;
        dy = (dy + 1.f) / 2.f;
        dz = (dz + 1.f) / 2.f;
      }
      float sh[16];
      sh_deg4(dx, dy, dz, sh);
#pragma unroll
      for (int k = 0; k < 16; ++k) CIN[k * LDA + s] = sh[k];
    }
    {
      const float* a = A.app_per_camera ? A.p.emb + cam_row * 32 : A.app_mean;
      CIN[(31 + 2 * lvl) * LDA + s] = a ? a[2 * lvl] : 0.f;
      CIN[(32 + 2 * lvl) * LDA + s] = a ? a[2 * lvl + 1] : 0.f;
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (have_prev) {  // the previous tile's hash-table gradient; its position gradient: the per-level partials are in A1
    if (!(A.debug_skip & 1)) {
      float gpx = 0.f, gpy = 0.f, gpz = 0.f;  // (unused: the <false> forms do not touch them)
      // (a wave holds two levels, 32 lanes each: the branch below splits it along whole 16-lane rows, which is all the
      //  DPP run-length reduction and the quad rounds reach across)
      if (lvl < A.cells.num_levels) {
        // cell-major level: the 16 sums of a run go through an LDS buffer per wave.  A2 / D2 / D1 / DCIN (2304 floats each)
        // are dead from the end of a tile to the s2 phase of the next, three barriers after this point: two waves each.
        float* tb = (wave < 2 ? A2 : wave < 4 ? D2 : wave < 6 ? D1 : DCIN) + (wave & 1) * (64 * 17);
        const unsigned nl = cell_n_of(A.cells, lvl);
        float* rec = A.cells.base + cell_offset_of(A.cells, lvl) +
                     (size_t)(blockIdx.x % cell_copies_of(A.cells, lvl)) * ((size_t)nl * nl * nl * 16);
        hash_level_backward_cells_rows(rec, nl, tb, A.g.table, A.p.table, my_lv, A.grid.pos_offset, p_px, p_py, p_pz, p_g0, p_g1,
                                       lane);
      } else if (lvl == 0 && A.coarse.base) {
        float* mine = A.coarse.base + (size_t)(blockIdx.x % A.coarse.copies) * (2u * A.coarse.n1 * A.coarse.n1 * A.coarse.n1);
        hash_level_backward_private<false>(mine, A.coarse.n1, A.g.table, A.p.table, my_lv, A.grid.pos_offset, p_px, p_py,
                                           p_pz, p_g0, p_g1, lane, gpx, gpy, gpz);
      } else
        hash_level_backward<false>(A.g.table, A.p.table, my_lv, A.grid.pos_offset, p_px, p_py, p_pz, p_g0, p_g1, lane, gpx,
                                   gpy, gpz);
    }
    if (A.d_pos) {
      // one thread per sample sums the 16 levels' partials, written at the end of the previous trip (A1 is not written before
      // the s1 phase, two barriers away)
      const float* part = A1;
      if (lvl == 0 && p_valid) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
          gx += part[(3 * l + 0) * LDA + s];
          gy += part[(3 * l + 1) * LDA + s];
          gz += part[(3 * l + 2) * LDA + s];
        }
        normalize_position_backward(A.scene, p_wx, p_wy, p_wz, p_self, gx, gy, gz);
        A.d_pos[3 * p_ismp] = gx;
        A.d_pos[3 * p_ismp + 1] = gy;
        A.d_pos[3 * p_ismp + 2] = gz;
      }
    }
    }
    if (!live) break;
    __builtin_amdgcn_sched_barrier(0);
    // ---- h1 = relu(W0 enc + b0) -----------------------------------------------------------------------------------------
    store_blk<true>(H1, n0, s0, blk_fwd<32, MM>(Wb0, 36, n0, ENC, s0, bias4(lds + B_0, n0, lane), lane), lane);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- o16 = W1 h1 + b1; geo rows also into the colour input ---------------------------------------------------------
    if (wave < 2) {
      const int c0 = 16 * wave;
      const f32x4 v = blk_fwd<64, MM>(Wb1, 68, 0, H1, c0, bias4(lds + B_1, 0, lane), lane);
      CN_LANE_IQ
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * q + rr;
        O16[row * LDA + c0 + i] = v[rr];
        if (row > 0) CIN[(15 + row) * LDA + c0 + i] = v[rr];
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- semantic branch (its input is the DETACHED geo: nothing flows back to the base MLP) ---------------------------
    if (!(A.debug_skip & 32)) {  // (bit 32, timing only: the semantic branch's four phases skipped)
    store_blk<true>(A1, n0, s0, blk_fwd<16, MM>(Ws0, 20, n0, O16 + LDA, s0, bias4(lds + B_S0, n0, lane), lane), lane);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_blk<false>(A2, n0, s0, blk_fwd<64, MM>(Ws1, 68, n0, A1, s0, bias4(lds + B_S1, n0, lane), lane), lane);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_delta(D2, nullptr, n0, s0, blk_bwd<16, MM>(Wsem, 68, n0, DSEM, s0, zero4, lane), bS1, lane);  // d_s2
    if (wave >= 4) gY = blk_dw<MM>(DSEM, 0, A2, 16 * (wave - 4), gY, lane);                             // dW head
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_delta(D1, A1, n0, s0, blk_bwd<64, MM>(Ws1, 68, n0, D2, s0, zero4, lane), bS0, lane);          // d_s1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int blk = 2 * wave + b;
      __builtin_amdgcn_sched_barrier(0);
      gS1[b] = blk_dw<MM>(D2, 16 * (blk >> 2), A1, 16 * (blk & 3), gS1[b], lane);                        // dW sem1
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    }
    // ---- colour branch -----------------------------------------------------------------------------------------------------
    if (wave < 4) gX = blk_dw<MM>(D1, 16 * wave, O16 + LDA, 0, gX, lane);                               // dW sem0
    store_blk<true>(A1, n0, s0, blk_fwd<64, MM>(Wc0, 68, n0, CIN, s0, bias4(lds + B_C0, n0, lane), lane), lane);  // c1
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_blk<true>(A2, n0, s0, blk_fwd<64, MM>(Wc1, 68, n0, A1, s0, bias4(lds + B_C1, n0, lane), lane), lane);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 2) {  // rgb = sigmoid(Wc2 c2 + bc2); delta_pre = d_rgb * rgb * (1 - rgb)
      const int c0 = 16 * wave;
      const f32x4 v = blk_fwd<64, MM>(Wrgb, 68, 0, A2, c0, bias4(lds + B_RGB, 0, lane), lane);
      CN_LANE_IQ
      if (q == 0) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const float sg = 1.f / (1.f + expf(-v[rr]));
          const float d = drgb_in[rr] * sg * (1.f - sg);  // (zero past the end of the batch)
          DRGB[rr * LDA + c0 + i] = d;
          b_rgb[rr] += d;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_delta(D2, A2, n0, s0, blk_bwd<16, MM>(Wrgb, 68, n0, DRGB, s0, zero4, lane), bC1, lane);       // d_c2
    if (wave < 4) gY = blk_dw<MM>(DRGB, 0, A2, 16 * wave, gY, lane);                                    // dW rgb head
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_delta(D1, A1, n0, s0, blk_bwd<64, MM>(Wc1, 68, n0, D2, s0, zero4, lane), bC0, lane);          // d_c1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int blk = 2 * wave + b;
      __builtin_amdgcn_sched_barrier(0);
      gC1[b] = blk_dw<MM>(D2, 16 * (blk >> 2), A1, 16 * (blk & 3), gC1[b], lane);                        // dW col1
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_blk<false>(DCIN, n0, s0, blk_bwd<64, MM>(Wc0, 68, n0, D1, s0, zero4, lane), lane);            // d_cin
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int blk = 2 * wave + b;
      __builtin_amdgcn_sched_barrier(0);
      gC0[b] = blk_dw<MM>(D1, 16 * (blk >> 2), CIN, 16 * (blk & 3), gC0[b], lane);                       // dW col0
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- d_o16 (row 0: density logit through trunc_exp and the selector; rows 1..15: geo from the colour branch) -----
    {
      const int row = lvl;  // 16 rows x 32 samples = one entry per thread
      float v;
      if (row == 0) {
        const float logit = O16[s];
        v = dd_in * self * expf(fminf(fmaxf(logit, -15.f), 15.f));
      } else {
        v = DCIN[(15 + row) * LDA + s];
      }
      DO16[row * LDA + s] = v;
      b_o16 += v;
      // appearance-embedding gradient: rows 31..62, two per thread.  A tile's samples belong to one ray or to a few
      // consecutive ones (one camera row each): sum the runs of equal ray inside every 16-lane row first and let the last
      // lane of a run add the two sums -- per-sample atomics on a straddling tile were 1024 same-address requests per tile
      // (3.1 of the kernel's 19.2 ms at 65 536 rays x 48 samples, where every third tile straddles two rays).
      if (A.app_per_camera && !(A.debug_skip & 2)) {
        float g0 = valid ? DCIN[(31 + 2 * lvl) * LDA + s] : 0.f;
        float g1 = valid ? DCIN[(32 + 2 * lvl) * LDA + s] : 0.f;
        const bool last = row_run_reduce(valid ? (unsigned)r : 0xffffffffu, g0, g1, lane & 15);
        if (last && valid) {
          float* ge = A.g.emb + cam_row * 32 + 2 * lvl;
          if (g0 != 0.f) cn_atomic_add(ge, g0);
          if (g1 != 0.f) cn_atomic_add(ge + 1, g1);
        }
      }
      if (A.d_dir && lvl == 2 && valid) {
        float gsh[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) gsh[k] = DCIN[k * LDA + s];
        float dx = dirx, dy = diry, dz = dirz;
        const float chain = A.sh_unit ? 1.f : 0.5f;
        if (!A.sh_unit) {
          dx = (dx + 1.f) / 2.f;
          dy = (dy + 1.f) / 2.f;
          dz = (dz + 1.f) / 2.f;
        }
        float gx, gy, gz;
        sh_deg4_backward(dx, dy, dz, gsh, gx, gy, gz);
        A.d_dir[3 * ismp] = gx * chain;
        A.d_dir[3 * ismp + 1] = gy * chain;
        A.d_dir[3 * ismp + 2] = gz * chain;
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- base MLP backward ------------------------------------------------------------------------------------------------
    store_delta(D2, H1, n0, s0, blk_bwd<16, MM>(Wb1, 68, n0, DO16, s0, zero4, lane), bH1, lane);        // d_h1
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 4) {
      store_blk<false>(D1, 16 * (wave >> 1), 16 * (wave & 1),
                       blk_bwd<64, MM>(Wb0, 36, 16 * (wave >> 1), D2, 16 * (wave & 1), zero4, lane), lane);  // d_enc
    } else {
      gX = blk_dw<MM>(DO16, 0, H1, 16 * (wave - 4), gX, lane) /* dW base1 */;
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    gB0 = blk_dw<MM>(D2, 16 * (wave >> 1), ENC, 16 * (wave & 1), gB0, lane);                            // dW base0
    {
      const bool lvl_off = (A.debug_skip >> (8 + lvl)) & 1;  // bits 8..23: skip the scatter of level l (profiling)
      p_g0 = valid && !lvl_off ? D1[(2 * lvl) * LDA + s] : 0.f;
      p_g1 = valid && !lvl_off ? D1[(2 * lvl + 1) * LDA + s] : 0.f;
      if (A.d_pos) {  // this (sample, level)'s share of d(loss)/d(normalised position); A1 is dead since the d_c1 phase
        const bool pos_off = (A.debug_skip & 1) != 0;
        A1[(3 * lvl + 0) * LDA + s] = pos_off ? 0.f : p_g0 * jx.x + p_g1 * jx.y;
        A1[(3 * lvl + 1) * LDA + s] = pos_off ? 0.f : p_g0 * jy.x + p_g1 * jy.y;
        A1[(3 * lvl + 2) * LDA + s] = pos_off ? 0.f : p_g0 * jz.x + p_g1 * jz.y;
      }
      p_px = px;
      p_py = py;
      p_pz = pz;
      p_wx = wx;
      p_wy = wy;
      p_wz = wz;
      p_self = self;
      p_ismp = ismp;
      p_valid = valid;
      have_prev = true;
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();

  // ---- flush -------------------------------------------------------------------------------------------------------------
  if (!(A.debug_skip & 4)) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int blk = 2 * wave + b;
      __builtin_amdgcn_sched_barrier(0);
      flush_dw(A.g.ws1, 64, 64, 16 * (blk >> 2), 16 * (blk & 3), gS1[b], lane);
      flush_dw(A.g.wc1, 64, 64, 16 * (blk >> 2), 16 * (blk & 3), gC1[b], lane);
      flush_dw(A.g.wc0, 64, 63, 16 * (blk >> 2), 16 * (blk & 3), gC0[b], lane);
    }
    flush_dw(A.g.w0, 64, 32, 16 * (wave >> 1), 16 * (wave & 1), gB0, lane);
    if (wave < 4) {
      flush_dw(A.g.ws0, 64, 15, 16 * wave, 0, gX, lane);
      flush_dw(A.g.wc2, 3, 64, 0, 16 * wave, gY, lane);
    } else {
      flush_dw(A.g.w1, 16, 64, 0, 16 * (wave - 4), gX, lane);
      flush_dw(A.g.wh, 1, 64, 0, 16 * (wave - 4), gY, lane);
    }
  }
  flush_bias(A.g.bs1, n0, bS1, lane);
  flush_bias(A.g.bs0, n0, bS0, lane);
  flush_bias(A.g.bc1, n0, bC1, lane);
  flush_bias(A.g.bc0, n0, bC0, lane);
  flush_bias(A.g.b0, n0, bH1, lane);
  // per-thread scalars: d_o16 rows (thread's row = lvl), rgb head (waves 0,1 lanes q == 0), semantic head (lvl == 0)
  {
    float v = row16_sum(b_o16);
    const float other = __shfl_up(v, 16, 32);
    if (s == 31 && (v + other) != 0.f) cn_atomic_add(A.g.b1 + lvl, v + other);
    v = row16_sum(b_sem);
    const float o2 = __shfl_up(v, 16, 32);
    if (lvl == 0 && s == 31 && (v + o2) != 0.f) cn_atomic_add(A.g.bh, v + o2);
    if (wave < 2) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const float t = row16_sum(b_rgb[rr]);
        if (lane == 15 && t != 0.f) cn_atomic_add(A.g.bc2 + rr, t);
      }
    }
  }
}
#undef CN_MFMA

}  // namespace mf
}  // namespace cn

// cn_proposal_backward, round-4 form: ONE WAVE per 64-sample tile (included by train_field.hip after the helpers it uses).
//
// The first form (proposal_backward_kernel above: four waves share a tile, levels dealt out over the waves, activations in
// LDS, six workgroup barriers per tile) ran at half its atomic-request floor: 2.8 ms per launch at 65 536 rays = 1.2 ms
// scatter + 0.4 ms weight-gradient dots on the VALU + 1.2 ms of gathers / tiny layers / barriers, one after the other -- per
// tile a workgroup waits for the slowest wave (the finest level's scatter) five times, wave 0 alone computes positions and the
// logit, and only four tiles per CU are in flight.  A 2L -> 16 -> 1 network is small enough for a lane to carry its sample
// through the whole chain in registers:
//   * lane = sample: position, the L levels' gathers (issued level after level, no barrier between them), 2L -> 16 -> 1 forward,
//     the two deltas and d(enc) are per-lane FMAs on weights from scalar loads -- the same products in the same order as the
//     first form, so table gradients, d(position) and d(density) paths are bit-identical;
//   * weight gradients dW0[16][2L] = sum over samples delta_h x enc: ONE 16 x 16 MFMA accumulator per wave
//     (v_mfma_f32_16x16x4_f32, sixteen per tile), operands through two wave-private [16][68] LDS images; column 2L of the enc
//     image is 1, so the same accumulator's column 2L is the bias gradient.  dW1 / db1 are per-lane accumulators reduced once
//     at the end of the kernel;
//   * the scatter routines for the private copies and the table path are the first form's, called for every level by every wave;
//     the cell-major path (hash_level_backward_cells_rows) reduces runs after its transpose instead of before; its transpose
//     buffer aliases the delta_h image;
//   * no __syncthreads in the tile loop: sixteen independent waves per CU, each with its own tile, overlap each other's gather
//     latency, matrix work and atomics.  One barrier at the end folds the four waves' weight gradients before the atomics.
#pragma once

namespace cn {
namespace pw {

constexpr int XS = 68;                  // row stride of an operand image (floats): b128 reads of 16 rows hit 64 distinct banks
// A wave's LDS (floats): the delta_h image [16][XS] at 0, the enc image [16][XS] at XENC.  Two other tenants alias them: the
// [64][17] transpose buffer of the cell-major scatter (exactly the delta_h image) and, from the gathers to the position
// gradient, the levels' Jacobians [6 L][64] -- which must end below the enc image's constant rows (row 2L = 1, rows above = 0).
template <int L>
struct WaveLds {
  static constexpr int JAC = 6 * L * 64;
  static constexpr int XENC = (JAC - 2 * L * XS > 16 * XS ? JAC - 2 * L * XS : 16 * XS + 3) / 4 * 4;
  static constexpr int SIZE = XENC + 16 * XS;
  static_assert(16 * XS >= 64 * 17, "the transpose buffer must fit the delta_h image");
  static_assert(XENC + 2 * L * XS >= JAC && XENC >= 16 * XS && XENC % 4 == 0, "layout");
};
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sums over the 64 lanes of sixteen per-lane values at once, by halving: a swap of register halves + an add folds two values
// into one register (v_permlane32_swap, then v_permlane16_swap), three DPP steps fold four registers of 16-lane rows into one,
// two quad steps finish.  Returns, in every lane, the total of value VALUE_OF_LANE(lane) = 4 p[(lane >> 2) & 3] + p[lane >> 4],
// p = {0, 2, 1, 3}.  35 instructions instead of 16 scans.
__device__ __forceinline__ float sum16_over_lanes(const float (&v)[16], int lane) {
  float u[8], w[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned a = __builtin_bit_cast(unsigned, v[2 * j]), b = __builtin_bit_cast(unsigned, v[2 * j + 1]);
    permlane32_swap(a, b);
    u[j] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);  // lanes 0..31: value 2j, lanes 32..63: value 2j + 1
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    unsigned a = __builtin_bit_cast(unsigned, u[2 * m]), b = __builtin_bit_cast(unsigned, u[2 * m + 1]);
    permlane16_swap(a, b);
    w[m] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);  // rows: values 4m, 4m + 2, 4m + 1, 4m + 3
  }
  const bool b8 = (lane & 8) != 0, b4 = (lane & 4) != 0;
  float t[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float x = b8 ? w[2 * h + 1] : w[2 * h], y = b8 ? w[2 * h] : w[2 * h + 1];
    t[h] = x + dpp_f32<0x140>(y);  // row_mirror: lanes 0..7 of a row fold w[2h], lanes 8..15 fold w[2h + 1]
  }
  const float x = b4 ? t[1] : t[0], y = b4 ? t[0] : t[1];
  float z = x + dpp_f32<0x141>(y);  // row_half_mirror: quads 0 / 2 fold t[0] (w[0] / w[1]), quads 1 / 3 fold t[1] (w[2] / w[3])
  z += dpp_f32<0xB1>(z);            // quad_perm [1,0,3,2]
  z += dpp_f32<0x4E>(z);            // quad_perm [2,3,0,1]
  return z;
}
__device__ __forceinline__ int sum16_value_of_lane(int lane) {
  const int g = (lane >> 2) & 3, r = lane >> 4;
  const int pg = ((g & 1) << 1) | (g >> 1), pr = ((r & 1) << 1) | (r >> 1);  // p = {0, 2, 1, 3}: the two bits swapped
  return 4 * pg + pr;
}

template <int L>
__global__ void __launch_bounds__(256, 4) proposal_backward_wave_kernel(PropBwdArgs A) {
  constexpr int K = 2 * L, H = 16;
  static_assert(K + 1 <= 16, "enc columns + the bias column must fit one 16-wide MFMA block");
  typedef WaveLds<L> WL;
  __shared__ __attribute__((aligned(16))) float lds[4 * WL::SIZE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* xdh = lds + wave * WL::SIZE;  // [16][XS]: delta_h[n][sample]
  float* xenc = xdh + WL::XENC;        // [16][XS]: enc[k][sample], row K = 1, rows above stay 0
  float* tb = xdh;
  float* jac = xdh;                    // [6 L][64]
  const int i16 = lane & 15, q = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // dW0 | db0: D[n = 4 q + r][k = i16]
  float gw1 = 0.f, gb1 = 0.f;  // dW1[sum16_value_of_lane(lane)] (every tile's sum over the lanes), db1 (this lane's samples)
  // constant rows of the enc image (the transpose buffer covers exactly the delta_h image): row K = 1, the rows above it 0
  xenc[K * XS + lane] = 1.f;
#pragma unroll
  for (int k = K + 1; k < 16; ++k) xenc[k * XS + lane] = 0.f;
  const long long total = A.R * (long long)A.S;
  const long long ntiles = (total + TS - 1) / TS;
  const bool small = total < (1ll << 31);  // (32-bit ray index division where it is exact)
  // one contiguous run of tiles per wave (see field_backward_mfma_kernel: batches sorted by camera and pixel)
  const long long nwaves = gridDim.x * 4ll, gw = blockIdx.x * 4ll + wave, per = (ntiles + nwaves - 1) / nwaves;
  const long long t_end = (gw + 1) * per < ntiles ? (gw + 1) * per : ntiles;
  for (long long tile = gw * per; tile < t_end; ++tile) {
    const long long i = tile * TS + lane;
    const bool valid = i < total;
    const long long ic = valid ? i : total - 1;
    const long long r = small ? (long long)((unsigned)ic / (unsigned)A.S) : ic / A.S;
    const float mid = (A.starts[ic] + A.ends[ic]) / 2.f;
    const float wx = A.origins[3 * r] + A.directions[3 * r] * mid;
    const float wy = A.origins[3 * r + 1] + A.directions[3 * r + 1] * mid;
    const float wz = A.origins[3 * r + 2] + A.directions[3 * r + 2] * mid;
    float px = wx, py = wy, pz = wz;
    const float sel_f = normalize_position(A.scene, px, py, pz) ? 1.f : 0.f;
    // ---- encoding: every level's features and their Jacobian with respect to the position ------------------------------------
    // (the Jacobians wait in LDS until the deltas are known: thirty registers otherwise.  LDS operations of one wave execute
    //  in order, so the previous tile's scatter has read its transpose buffer by now)
    float enc[K];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      v2f_t jx, jy, jz;
      const float2 f = hash_level_jac(A.table, A.grid.level(l), A.grid.pos_offset, px, py, pz, jx, jy, jz);
      enc[2 * l] = f.x;
      enc[2 * l + 1] = f.y;
      jac[(6 * l + 0) * 64 + lane] = jx.x;
      jac[(6 * l + 1) * 64 + lane] = jx.y;
      jac[(6 * l + 2) * 64 + lane] = jy.x;
      jac[(6 * l + 3) * 64 + lane] = jy.y;
      jac[(6 * l + 4) * 64 + lane] = jz.x;
      jac[(6 * l + 5) * 64 + lane] = jz.y;
      // (pinned: otherwise hipcc issues every level's 8 gathers before the first blend -- 80 registers of corner values)
      asm volatile("" : "+v"(enc[2 * l]), "+v"(enc[2 * l + 1]));
    }
    // ---- 2L -> 16 (ReLU) -> 1, trunc_exp; the deltas ----------------------------------------------------------------------------
    // (the first-layer weights are read in two places; each gets its own laundered pointer, so that the compiler re-issues the
    //  scalar loads instead of keeping 16 x 2L values alive in -- spilled -- SGPRs from one use to the other)
    cfloat_ptr W0 = as_const(A.w0), B0 = as_const(A.b0), W1 = as_const(A.w1), B1 = as_const(A.b1);
    asm volatile("" : "+s"(W0), "+s"(B0), "+s"(W1), "+s"(B1));
    float hid[H];
#pragma unroll
    for (int n = 0; n < H; ++n) {
      float a = B0[n];
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(W0[n * K + k], enc[k], a);
      hid[n] = fmaxf(a, 0.f);
    }
    float logit = B1[0];
#pragma unroll
    for (int n = 0; n < H; ++n) logit = fmaf(W1[n], hid[n], logit);
    const float up = valid ? A.d_density[ic] : 0.f;
    const float dout = up * sel_f * expf(fminf(fmaxf(logit, -15.f), 15.f));
    gb1 += dout;
    float dh[H];
    {
      float t[H];
#pragma unroll
      for (int n = 0; n < H; ++n) {
        t[n] = dout * hid[n];
        dh[n] = hid[n] > 0.f ? W1[n] * dout : 0.f;
      }
      gw1 += sum16_over_lanes(t, lane);
    }
    // ---- delta_enc[k] = sum_n W0[n][k] delta_h[n]; d(position) from the levels' Jacobians ----------------------------------------
    cfloat_ptr W0t = as_const(A.w0);
    asm volatile("" : "+s"(W0t));
    float denc[K];
    float gpx = 0.f, gpy = 0.f, gpz = 0.f;
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      float g0 = 0.f, g1 = 0.f;
#pragma unroll
      for (int n = 0; n < H; ++n) {
        g0 = fmaf(W0t[n * K + 2 * l], dh[n], g0);
        g1 = fmaf(W0t[n * K + 2 * l + 1], dh[n], g1);
      }
      g0 = valid ? g0 : 0.f;
      g1 = valid ? g1 : 0.f;
      denc[2 * l] = g0;
      denc[2 * l + 1] = g1;
      gpx += g0 * jac[(6 * l + 0) * 64 + lane] + g1 * jac[(6 * l + 1) * 64 + lane];
      gpy += g0 * jac[(6 * l + 2) * 64 + lane] + g1 * jac[(6 * l + 3) * 64 + lane];
      gpz += g0 * jac[(6 * l + 4) * 64 + lane] + g1 * jac[(6 * l + 5) * 64 + lane];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dW0 | db0 += delta_h^T [enc | 1] on the matrix pipe ----------------------------------------------------------------------
#pragma unroll
    for (int n = 0; n < H; ++n) xdh[n * XS + lane] = dh[n];
#pragma unroll
    for (int k = 0; k < K; ++k) xenc[k * XS + lane] = enc[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!(A.debug_skip & 16)) {
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xdh + i16 * XS + 16 * sb + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xenc + i16 * XS + 16 * sb + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = a[e], bv = b[e];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- delta_enc straight into the table gradient, level by level (fine levels first: their requests are the many) ----------
    // (the position is laundered: with its provenance visible hipcc keeps every level's cell coordinates and weights of the
    //  gather phase alive across the network for the scatter -- 45 registers -- instead of recomputing them)
    asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      if ((A.debug_skip & 8) || ((A.debug_skip >> (8 + l)) & 1)) continue;  // profiling: skip the scatter (of level l)
      const float g0 = denc[2 * l], g1 = denc[2 * l + 1];
      float ux = 0.f, uy = 0.f, uz = 0.f;  // (unused: the <false> forms do not touch them)
      if (l < A.cells.num_levels) {
        const unsigned nl = A.cells.n[l];
        float* rec = A.cells.base + A.cells.offset[l] + (size_t)(blockIdx.x % A.cells.copies[l]) * ((size_t)nl * nl * nl * 16);
        hash_level_backward_cells_rows(rec, nl, tb, A.g_table, A.table, A.grid.level(l), A.grid.pos_offset, px, py, pz, g0, g1, lane);
      } else if (l == 0 && A.coarse.base) {
        float* mine = A.coarse.base + (size_t)(blockIdx.x % A.coarse.copies) * (2u * A.coarse.n1 * A.coarse.n1 * A.coarse.n1);
        hash_level_backward_private<false>(mine, A.coarse.n1, A.g_table, A.table, A.grid.level(0), A.grid.pos_offset, px, py, pz,
                                           g0, g1, lane, ux, uy, uz);
      } else
        hash_level_backward<false>(A.g_table, A.table, A.grid.level(l), A.grid.pos_offset, px, py, pz, g0, g1, lane, ux, uy, uz);
    }
    if (A.d_pos && valid) {
      normalize_position_backward(A.scene, wx, wy, wz, sel_f, gpx, gpy, gpz);
      A.d_pos[3 * i] = gpx;
      A.d_pos[3 * i + 1] = gpy;
      A.d_pos[3 * i + 2] = gpz;
    }
  }
  // ---- the four waves' weight gradients: one sum through LDS, one atomic per entry and workgroup ----------------------------------
  __syncthreads();
  float* red = lds;             // [4][256]
  float* red1 = lds + 4 * 256;  // [4][17]
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) red[wave * 256 + (4 * q + rr) * 16 + i16] = acc[rr];
  if ((lane & 3) == 0) red1[wave * 17 + sum16_value_of_lane(lane)] = gw1;
  {
    const float s = wave_sum(gb1);
    if (lane == 0) red1[wave * 17 + 16] = s;
  }
  __syncthreads();
  {
    const float s = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
    const int n = tid >> 4, k = tid & 15;
    if (s != 0.f) {
      if (k < K) cn_atomic_add(A.g_w0 + n * K + k, s);
      else if (k == K) cn_atomic_add(A.g_b0 + n, s);
    }
  }
  if (tid < 17) {
    const float s = red1[tid] + red1[17 + tid] + red1[34 + tid] + red1[51 + tid];
    if (s != 0.f) cn_atomic_add(tid < 16 ? A.g_w1 + tid : A.g_b1, s);
  }
}

}  // namespace pw
}  // namespace cn

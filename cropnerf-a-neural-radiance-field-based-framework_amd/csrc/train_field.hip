// Training, field side: parameter gradients of FruitField and of the proposal HashMLPDensityFields.
//
//   cn_field_backward     : given d loss / d (density, rgb, semantics) per sample (train_render.hip), recompute the
//                           forward of FruitField (fruit_nerf/fruit_field.py:169-282, training branch: per-camera
//                           appearance, semantic MLP on detached geo features) for a 64-sample tile with all activations
//                           in LDS, back-propagate, and accumulate gradients of every Linear layer, of the appearance
//                           embedding and of the hash table.
//   cn_proposal_backward  : same for a proposal network given d loss / d density (interlevel loss).
//
// Layout: a 256-thread workgroup owns a tile of 64 samples; activations and deltas live in LDS as [feature][65]
// (row pad 1 -> conflict-free both for "lane = sample" sweeps and for the weight-gradient dots).  Forward / delta
// layers split their output rows over the 4 waves (weights come through scalar loads).  Weight gradients
// dW[n][k] = sum_samples delta[n] x[k]: thread t owns entries t, t+256, ... of each matrix and keeps the partial sums in
// registers across all tiles of its (persistent) workgroup, then issues one global_atomic_add_f32 per entry at the end
// -- atomics per step are (#workgroups x #parameters), not (#samples x #parameters).  Hash-table gradients are
// scatter-adds (2 floats x 8 corners per level per sample), as in every hash-grid trainer.
//
// Shapes: the default fruit_nerf_method field (16 levels, 32->64->16, 15->64->64->1, 63->64->64->3, appearance 32)
// and {5|7}-level 2L->16->1 proposal nets; other shapes return CN_ERR_UNSUPPORTED.
#include <algorithm>
#include <cstring>
#include <cstdlib>

#include <mutex>

#include "cn_common.hpp"
#include "cn_det.hpp"
#include "wave_ops.hpp"

namespace cn {

constexpr int TS = 64;       // samples per tile
constexpr int LD = TS + 1;   // padded row length
constexpr int TB = 256;      // threads per workgroup

// Weights are read-only for the whole launch (gradients go to separate buffers), so they are addressed through the
// constant address space: with a wave-uniform index the compiler then emits s_load (SGPR operands for v_fma) instead of
// one broadcast global_load per multiply -- it cannot prove invariance for plain global pointers next to the atomics.
typedef const float __attribute__((address_space(4)))* cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_const(const float* p) { return (cfloat_ptr)(unsigned long long)p; }

// y[n][lane] = act(b[n] + sum_k W[n][k] x[k][lane]) for the rows n = wave, wave+4, ...
template <int K, int N, bool RELU>
__device__ __forceinline__ void fwd_rows(const float* __restrict__ Wg, const float* __restrict__ bg, const float* x,
                                         float* y, int wave, int lane) {
  const cfloat_ptr W = as_const(Wg), b = as_const(bg);
  for (int n = wave; n < N; n += 4) {
    float acc = b[n];
#pragma unroll 8
    for (int k = 0; k < K; ++k) acc = fmaf(W[n * K + k], x[k * LD + lane], acc);
    y[n * LD + lane] = RELU ? fmaxf(acc, 0.f) : acc;
  }
}

// dx[k][lane] = (sum_n W[n][k] dy[n][lane]) * (gate ? act[k][lane] > 0 : 1) for rows k = k0 + wave, +4, ... < k1
template <int K, int N>
__device__ __forceinline__ void bwd_rows(const float* __restrict__ Wg, const float* dy, float* dx, const float* act,
                                         int k0, int k1, int wave, int lane) {
  const cfloat_ptr W = as_const(Wg);
  for (int k = k0 + wave; k < k1; k += 4) {
    float acc = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) acc = fmaf(W[n * K + k], dy[n * LD + lane], acc);
    if (act) acc = act[k * LD + lane] > 0.f ? acc : 0.f;
    dx[k * LD + lane] = acc;
  }
}

// acc[i] += sum_j dy[n][j] x[k][j] for the entries e = tid + TB*i (n = e / K, k = e % K)
template <int K, int N>
struct WGrad {
  static constexpr int E = (N * K + TB - 1) / TB;
  float acc[E];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < E; ++i) acc[i] = 0.f;
  }
  __device__ __forceinline__ void add(const float* dy, const float* x, int tid) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      int e = tid + TB * i;
      if (e < N * K) {
        const float* a = dy + (e / K) * LD;
        const float* b = x + (e % K) * LD;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < TS; ++j) s = fmaf(a[j], b[j], s);
        acc[i] += s;
      }
    }
  }
  __device__ __forceinline__ void flush(float* __restrict__ g, int tid) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      int e = tid + TB * i;
      if (e < N * K) cn_atomic_add(g + e, acc[i]);
    }
  }
};

// bias gradient: thread n < N owns sum_j dy[n][j]
template <int N>
__device__ __forceinline__ void bias_add(float& acc, const float* dy, int tid) {
  if (tid < N) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < TS; ++j) s += dy[tid * LD + j];
    acc += s;
  }
}

// Run-length pre-reduction of scatter-adds inside each 16-lane row.  Lanes are consecutive samples of a ray, so at the
// coarser levels neighbouring lanes hit the same grid cell: runs of equal `key` are summed with a segmented scan on DPP
// row shifts (no LDS, no address registers) and only the last lane of a run issues the atomic.  Must be called by all
// 64 lanes (pass zeros for lanes with nothing to add).  Returns true where the (summed) v0 / v1 are to be added.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ bool row_run_reduce(unsigned key, float& v0, float& v1, int row_lane) {
  const unsigned prev = dpp_u32<0x111>(key);  // row_shr:1
  unsigned head = (row_lane == 0 || prev != key) ? 1u : 0u;
  const unsigned next_head = dpp_u32<0x101>(head);  // row_shl:1 (0 past the row end)
  const bool last = row_lane == 15 || next_head != 0u;
  unsigned f = head;
#define CN_SEG_STEP(CTRL)                         \
  {                                               \
    const float a0 = dpp_f32<CTRL>(v0);           \
    const float a1 = dpp_f32<CTRL>(v1);           \
    const unsigned fu = dpp_u32<CTRL>(f);         \
    if (!f) {                                     \
      v0 += a0;                                   \
      v1 += a1;                                   \
      f |= fu;                                    \
    }                                             \
  }
  CN_SEG_STEP(0x111)
  CN_SEG_STEP(0x112)
  CN_SEG_STEP(0x114)
  CN_SEG_STEP(0x118)
#undef CN_SEG_STEP
  return last;
}

// scatter d(loss)/d(features of one level) into the table gradient with the forward's trilinear weights
// (all 64 lanes call it; g0 = g1 = 0 for lanes without a sample).  POS: also accumulate d(loss)/d(normalised position)
// -- the trilinear weights are linear in the in-cell offset, so d enc_f / d x = scale * sum_c (+-1) wy wz table[c].f
// (the path HashEncoding.pytorch_fwd's `offset = scaled - floor(scaled)` carries gradient through; it is what feeds
// the camera pose refinement).
template <bool POS>
__device__ __forceinline__ void hash_level_backward(float* __restrict__ gtab, const float* __restrict__ table,
                                                    const Lvl& lv, float pos_offset, float px, float py, float pz,
                                                    float g0, float g1, int lane, float& dpx, float& dpy, float& dpz) {
  const Cell cell = hash_cell(lv, pos_offset, px, py, pz);
  const float ox = cell.ox, oy = cell.oy, oz = cell.oz, scale = lv.scale;
  const unsigned mask = lv.mask, level_off = lv.off;
  unsigned hx[2] = {cell.hx0, cell.hx1};
  unsigned hy[2] = {cell.hy0, cell.hy1};
  unsigned hz[2] = {cell.hz0, cell.hz1};
  float wx[2] = {1.f - ox, ox}, wy[2] = {1.f - oy, oy}, wz[2] = {1.f - oz, oz};  // index 1 = ceil corner
  const int row_lane = lane & 15;
  float ax = 0.f, ay = 0.f, az = 0.f;
  // The index xors ix into the low bits (hashed and dense levels alike), so the two corners of an x-edge lie in one
  // aligned 64-byte segment of the table unless ix = 7 (mod 8) -- and the memory pipe takes everything ONE instruction sends to one 64-byte segment as ONE
  // atomic request, whatever the lanes (tools/atomic_microbench.hip: 21e9 requests/s, the bound of this kernel).  Each
  // atomic instruction therefore serves ONE x-edge of one source lane from FOUR adjacent lanes (entry = lane & 2 ? x1
  // corner : x0 corner, feature = lane & 1); the four source lanes of a quad take turns: 4 requests per sample and
  // level for 7 of 8 cells instead of 16 single floats.
#pragma unroll
  for (int bd = 0; bd < 4; ++bd) {
    const int b = bd & 1, d = bd >> 1;
    unsigned eu[2];
    float v0[2], v1[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float w = wx[a] * wy[b] * wz[d];
      const unsigned e = ((hx[a] ^ hy[b] ^ hz[d]) & mask) + level_off;
      if constexpr (POS) {
        const float2 t = hash_gather(table, e);
        const float tg = t.x * g0 + t.y * g1;
        ax += (a ? tg : -tg) * (wy[b] * wz[d]);
        ay += (b ? tg : -tg) * (wx[a] * wz[d]);
        az += (d ? tg : -tg) * (wx[a] * wy[b]);
      }
      v0[a] = w * g0;
      v1[a] = w * g1;
      const bool issue = row_run_reduce(e, v0[a], v1[a], row_lane);
      eu[a] = issue && (v0[a] != 0.f || v1[a] != 0.f) ? e : 0xffffffffu;
    }
    const int ql = lane & 3;
#define CN_QUAD_ROUND(CTRL)                                                                          \
  {                                                                                                  \
    const unsigned e0 = dpp_u32<CTRL>(eu[0]), e1 = dpp_u32<CTRL>(eu[1]);                             \
    const float a00 = dpp_f32<CTRL>(v0[0]), a01 = dpp_f32<CTRL>(v1[0]);                              \
    const float a10 = dpp_f32<CTRL>(v0[1]), a11 = dpp_f32<CTRL>(v1[1]);                              \
    const unsigned es = (ql & 2) ? e1 : e0;                                                          \
    const float val = (ql & 2) ? ((ql & 1) ? a11 : a10) : ((ql & 1) ? a01 : a00);                    \
    if (es != 0xffffffffu) cn_atomic_add(gtab + 2 * (size_t)es + (ql & 1), val);                         \
  }
    CN_QUAD_ROUND(0x00)  // quad_perm [0,0,0,0]
    CN_QUAD_ROUND(0x55)  // [1,1,1,1]
    CN_QUAD_ROUND(0xAA)  // [2,2,2,2]
    CN_QUAD_ROUND(0xFF)  // [3,3,3,3]
#undef CN_QUAD_ROUND
  }
  // (History, measured at 4096 rays: one atomic per float from the owning lane 2.73 ms; the two features of an entry from
  // two adjacent lanes of one instruction 1.73 ms; this x-edge form 1.30 ms.  The earlier forms were removed.)
  if constexpr (POS) {
    dpx = fmaf(ax, scale, dpx);
    dpy = fmaf(ay, scale, dpy);
    dpz = fmaf(az, scale, dpz);
  }
}

// The same for the level whose gradient is accumulated in private dense copies (CoarseScatter): the atomics go to
// `priv` (this workgroup's copy) at the vertex's dense index; lanes whose cell lies outside the copy's n1^3 vertices
// (positions outside [0, 1]: only without scene contraction) take the table path afterwards.  The position gradient reads
// the parameter table at the real entries as before.
template <bool POS>
__device__ __forceinline__ void hash_level_backward_private(float* __restrict__ priv, unsigned n1,
                                                            float* __restrict__ gtab, const float* __restrict__ table,
                                                            const Lvl& lv, float pos_offset, float px, float py, float pz,
                                                            float g0, float g1, int lane, float& dpx, float& dpy,
                                                            float& dpz) {
  const Cell cell = hash_cell(lv, pos_offset, px, py, pz);
  const float ox = cell.ox, oy = cell.oy, oz = cell.oz, scale = lv.scale;
  // the integer cell coordinates again (hash_cell keeps their index terms only)
  const unsigned ix = cell.hx0;
  const unsigned iy = (unsigned)(int)floorf(fmaf(py, lv.scale, pos_offset));
  const unsigned iz = (unsigned)(int)floorf(fmaf(pz, lv.scale, pos_offset));
  const bool inside = ix + 1u < n1 && iy + 1u < n1 && iz + 1u < n1;  // unsigned: negative coordinates are huge
  const float h0 = inside ? g0 : 0.f, h1 = inside ? g1 : 0.f;
  unsigned hx[2] = {cell.hx0, cell.hx1};
  unsigned hy[2] = {cell.hy0, cell.hy1};
  unsigned hz[2] = {cell.hz0, cell.hz1};
  float wx[2] = {1.f - ox, ox}, wy[2] = {1.f - oy, oy}, wz[2] = {1.f - oz, oz};
  const int row_lane = lane & 15;
  float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
  for (int bd = 0; bd < 4; ++bd) {
    const int b = bd & 1, d = bd >> 1;
    unsigned eu[2];
    float v0[2], v1[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float w = wx[a] * wy[b] * wz[d];
      const unsigned e = inside ? (ix + a) + n1 * ((iy + b) + n1 * (iz + d)) : 0xfffffffeu;
      if constexpr (POS) {
        const float2 t = hash_gather(table, ((hx[a] ^ hy[b] ^ hz[d]) & lv.mask) + lv.off);
        const float tg = t.x * h0 + t.y * h1;
        ax += (a ? tg : -tg) * (wy[b] * wz[d]);
        ay += (b ? tg : -tg) * (wx[a] * wz[d]);
        az += (d ? tg : -tg) * (wx[a] * wy[b]);
      }
      v0[a] = w * h0;
      v1[a] = w * h1;
      const bool issue = row_run_reduce(e, v0[a], v1[a], row_lane);
      eu[a] = issue && (v0[a] != 0.f || v1[a] != 0.f) ? e : 0xffffffffu;
    }
    const int ql = lane & 3;
#define CN_QUAD_ROUND(CTRL)                                                                          \
  {                                                                                                  \
    const unsigned e0 = dpp_u32<CTRL>(eu[0]), e1 = dpp_u32<CTRL>(eu[1]);                             \
    const float a00 = dpp_f32<CTRL>(v0[0]), a01 = dpp_f32<CTRL>(v1[0]);                              \
    const float a10 = dpp_f32<CTRL>(v0[1]), a11 = dpp_f32<CTRL>(v1[1]);                              \
    const unsigned es = (ql & 2) ? e1 : e0;                                                          \
    const float val = (ql & 2) ? ((ql & 1) ? a11 : a10) : ((ql & 1) ? a01 : a00);                    \
    if (es != 0xffffffffu) cn_atomic_add(priv + 2 * (size_t)es + (ql & 1), val);                         \
  }
    CN_QUAD_ROUND(0x00)
    CN_QUAD_ROUND(0x55)
    CN_QUAD_ROUND(0xAA)
    CN_QUAD_ROUND(0xFF)
#undef CN_QUAD_ROUND
  }
  if constexpr (POS) {
    dpx = fmaf(ax, scale, dpx);
    dpy = fmaf(ay, scale, dpy);
    dpz = fmaf(az, scale, dpz);
  }
  // cells outside the private copy (never with scene contraction): the plain path, for those lanes only
  if (__builtin_amdgcn_ballot_w64(!inside && (g0 != 0.f || g1 != 0.f)) != 0ull)
    hash_level_backward<POS>(gtab, table, lv, pos_offset, px, py, pz, inside ? 0.f : g0, inside ? 0.f : g1, lane, dpx, dpy,
                             dpz);
}

// The scatter of one CELL-MAJOR level (CellScatter): every sample adds the 16 weighted values of its cell -- 8 corners x 2
// features -- to the cell's 64-byte record with ONE request: runs of consecutive samples in the same cell are summed first
// (the same DPP run-length reduction, keyed by the cell), the 16 sums of a run end go through a wave-private LDS buffer
// `tb` ([64][17] floats) so that 16 lanes carry one record, and in round k the 16 lanes of every row add the record of the
// row's k-th sample.  Cells outside the n^3 array (positions outside [0, 1]: only without scene contraction) take the table
// path afterwards.  All 64 lanes of the wave must call.
template <bool POS>
__device__ __forceinline__ void hash_level_backward_cells(float* __restrict__ rec, unsigned n, float* __restrict__ tb,
                                                          float* __restrict__ gtab, const float* __restrict__ table,
                                                          const Lvl& lv, float pos_offset, float px, float py, float pz,
                                                          float g0, float g1, int lane, float& dpx, float& dpy,
                                                          float& dpz) {
  const Cell cell = hash_cell(lv, pos_offset, px, py, pz);
  const float ox = cell.ox, oy = cell.oy, oz = cell.oz;
  const unsigned ix = cell.hx0;
  const unsigned iy = (unsigned)(int)floorf(fmaf(py, lv.scale, pos_offset));
  const unsigned iz = (unsigned)(int)floorf(fmaf(pz, lv.scale, pos_offset));
  const bool inside = ix < n && iy < n && iz < n;  // unsigned: negative coordinates are huge
  const float h0 = inside ? g0 : 0.f, h1 = inside ? g1 : 0.f;
  const unsigned key = inside ? ix + n * (iy + n * iz) : 0xfffffffeu;
  const float wx[2] = {1.f - ox, ox}, wy[2] = {1.f - oy, oy}, wz[2] = {1.f - oz, oz};
  const int row_lane = lane & 15;
  bool last = false, any = false;
  float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int a = c & 1, b = (c >> 1) & 1, d = c >> 2;
    const float w = wx[a] * wy[b] * wz[d];
    if constexpr (POS) {
      const unsigned hx = a ? cell.hx1 : cell.hx0, hy = b ? cell.hy1 : cell.hy0, hz = d ? cell.hz1 : cell.hz0;
      const float2 t = hash_gather(table, ((hx ^ hy ^ hz) & lv.mask) + lv.off);
      const float tg = t.x * h0 + t.y * h1;
      ax += (a ? tg : -tg) * (wy[b] * wz[d]);
      ay += (b ? tg : -tg) * (wx[a] * wz[d]);
      az += (d ? tg : -tg) * (wx[a] * wy[b]);
    }
    float v0 = w * h0, v1 = w * h1;
    last = row_run_reduce(key, v0, v1, row_lane);
    any = any || v0 != 0.f || v1 != 0.f;
    tb[lane * 17 + 2 * c] = v0;
    tb[lane * 17 + 2 * c + 1] = v1;
  }
  tb[lane * 17 + 16] = __builtin_bit_cast(float, (last && any && inside) ? key : 0xffffffffu);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int row0 = lane & 48;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const unsigned cellk = __builtin_bit_cast(unsigned, tb[(row0 + k) * 17 + 16]);
    if (cellk != 0xffffffffu) cn_atomic_add(rec + (size_t)cellk * 16 + row_lane, tb[(row0 + k) * 17 + row_lane]);
  }
  __builtin_amdgcn_wave_barrier();
  if constexpr (POS) {
    dpx = fmaf(ax, lv.scale, dpx);
    dpy = fmaf(ay, lv.scale, dpy);
    dpz = fmaf(az, lv.scale, dpz);
  }
  if (__builtin_amdgcn_ballot_w64(!inside && (g0 != 0.f || g1 != 0.f)) != 0ull)
    hash_level_backward<POS>(gtab, table, lv, pos_offset, px, py, pz, inside ? 0.f : g0, inside ? 0.f : g1, lane, dpx, dpy,
                             dpz);
}

// hash_level_backward_cells with the run-length reduction done AFTER the transpose: every lane writes its 16 weighted values
// and its cell to the wave-private buffer unreduced; then the 16 lanes of a row walk the row's 16 samples in order, each lane
// summing one of the 16 record entries, and add the sum to the cell's record whenever the next sample lies in another cell.
// Same requests as the DPP form (one per run of samples in a cell), a third of its instructions: no segmented scans -- 16
// values x 4 steps of cross-lane moves -- only a running sum.  The sums of a run are taken in sample order instead of as a
// tree, so the last bit may differ from the first form's.  All 64 lanes must call; gradient only (no position gradient).
__device__ __forceinline__ void hash_level_backward_cells_rows(float* __restrict__ rec, unsigned n, float* __restrict__ tb,
                                                               float* __restrict__ gtab, const float* __restrict__ table,
                                                               const Lvl& lv, float pos_offset, float px, float py, float pz,
                                                               float g0, float g1, int lane) {
  const Cell cell = hash_cell(lv, pos_offset, px, py, pz);
  const float ox = cell.ox, oy = cell.oy, oz = cell.oz;
  const unsigned ix = cell.hx0;
  const unsigned iy = (unsigned)(int)floorf(fmaf(py, lv.scale, pos_offset));
  const unsigned iz = (unsigned)(int)floorf(fmaf(pz, lv.scale, pos_offset));
  const bool inside = ix < n && iy < n && iz < n;  // unsigned: negative coordinates are huge
  const float h0 = inside ? g0 : 0.f, h1 = inside ? g1 : 0.f;
  const float wx[2] = {1.f - ox, ox}, wy[2] = {1.f - oy, oy}, wz[2] = {1.f - oz, oz};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float w = wx[c & 1] * wy[(c >> 1) & 1] * wz[c >> 2];
    tb[lane * 17 + 2 * c] = w * h0;
    tb[lane * 17 + 2 * c + 1] = w * h1;
  }
  tb[lane * 17 + 16] = __builtin_bit_cast(float, inside ? ix + n * (iy + n * iz) : 0xffffffffu);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int row0 = lane & 48, row_lane = lane & 15;
  float acc = 0.f;
  unsigned cur = __builtin_bit_cast(unsigned, tb[row0 * 17 + 16]);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    acc += tb[(row0 + k) * 17 + row_lane];
    const unsigned next = k < 15 ? __builtin_bit_cast(unsigned, tb[(row0 + k + 1) * 17 + 16]) : 0xfffffffdu;
    if (next != cur) {  // (uniform within the row)
      if (cur != 0xffffffffu && acc != 0.f) cn_atomic_add(rec + (size_t)cur * 16 + row_lane, acc);
      acc = 0.f;
    }
    cur = next;
  }
  __builtin_amdgcn_wave_barrier();
  if (__builtin_amdgcn_ballot_w64(!inside && (g0 != 0.f || g1 != 0.f)) != 0ull) {
    float ux = 0.f, uy = 0.f, uz = 0.f;
    hash_level_backward<false>(gtab, table, lv, pos_offset, px, py, pz, inside ? 0.f : g0, inside ? 0.f : g1, lane, ux, uy, uz);
  }
}

// fold the cell-major levels into the gradient table and zero their touched records: one thread per (copy, cell) record,
// all levels in one launch (workgroups [first_block[l], first_block[l + 1]) belong to level l)
struct CellFoldArgs {
  CellScatter c;
  unsigned first_block[CN_CELL_LEVELS + 1];
  Lvl lv[CN_CELL_LEVELS];
};
__global__ void __launch_bounds__(256) cell_scatter_fold_kernel(CellFoldArgs F, float* __restrict__ gtab) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) l = (k < F.c.num_levels && blockIdx.x >= F.first_block[k]) ? k : l;  // block-uniform
  unsigned n = F.c.n[0], copies = F.c.copies[0], first = F.first_block[0];
  unsigned long long off = F.c.offset[0];
  Lvl lv = F.lv[0];
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) {
    const bool m = l == k;
    n = m ? F.c.n[k] : n;
    copies = m ? F.c.copies[k] : copies;
    first = m ? F.first_block[k] : first;
    off = m ? F.c.offset[k] : off;
    lv.off = m ? F.lv[k].off : lv.off;
    lv.mask = m ? F.lv[k].mask : lv.mask;
    lv.m1 = m ? F.lv[k].m1 : lv.m1;
    lv.m2 = m ? F.lv[k].m2 : lv.m2;
  }
  const unsigned long long cells = (unsigned long long)n * n * n;
  const unsigned long long i = (blockIdx.x - first) * 256ull + threadIdx.x;
  if (i >= cells * copies) return;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4* p = reinterpret_cast<f32x4*>(F.c.base + off + i * 16);
  const f32x4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
  const float v[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
  bool any = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) any = any || v[j] != 0.f;
  if (!any) return;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  p[0] = zero;
  p[1] = zero;
  p[2] = zero;
  p[3] = zero;
  const unsigned long long cidx = i % cells;
  const unsigned x = (unsigned)(cidx % n), y = (unsigned)((cidx / n) % n), z = (unsigned)(cidx / ((unsigned long long)n * n));
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (v[2 * c] == 0.f && v[2 * c + 1] == 0.f) continue;
    const unsigned e = (((x + (c & 1)) ^ ((y + ((c >> 1) & 1)) * lv.m1) ^ ((z + (c >> 2)) * lv.m2)) & lv.mask) + lv.off;
    cn_atomic_add(gtab + 2 * (size_t)e, v[2 * c]);
    cn_atomic_add(gtab + 2 * (size_t)e + 1, v[2 * c + 1]);
  }
}
// The same fold by BLOCKS of 8 x 8 x 8 cells (round 4).  The kernel above adds every touched record's 16 values to the table
// one float per atomic instruction: up to 16 requests per record and copy, 4.3e6 per call at 65 536 rays -- the fold was bound
// by its own atomics (0.27 ms per call, three calls per iteration).  Here a workgroup owns a block of cells in up to four of
// the level's copies, sums their records into the block's 9 x 9 x 9 vertices in LDS (ds_add_f32), and then adds every non-zero
// vertex to the table ONCE, two lanes per vertex (its two features) and vertices in x order: the index function xors x into
// the low bits, so the eight vertices of an aligned x-row of the block lie on one 64-byte line and travel as one request.
// Requests per block: ~2 per (y, z) row of vertices instead of up to 16 per record.
struct CellFoldBlocksArgs {
  CellScatter c;
  unsigned first_block[CN_CELL_LEVELS + 1];
  unsigned nb[CN_CELL_LEVELS];      // blocks per axis
  unsigned groups[CN_CELL_LEVELS];  // workgroups per block: the level's copies are dealt out over them
  Lvl lv[CN_CELL_LEVELS];
};
__global__ void __launch_bounds__(256) cell_scatter_fold_blocks_kernel(CellFoldBlocksArgs F, float* __restrict__ gtab) {
  __shared__ float acc[2 * 729];
  int l = 0;
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) l = (k < F.c.num_levels && blockIdx.x >= F.first_block[k]) ? k : l;  // block-uniform
  unsigned n = F.c.n[0], copies = F.c.copies[0], first = F.first_block[0], nb = F.nb[0], groups = F.groups[0];
  unsigned long long off = F.c.offset[0];
  Lvl lv = F.lv[0];
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) {
    const bool m = l == k;
    n = m ? F.c.n[k] : n;
    copies = m ? F.c.copies[k] : copies;
    first = m ? F.first_block[k] : first;
    nb = m ? F.nb[k] : nb;
    groups = m ? F.groups[k] : groups;
    off = m ? F.c.offset[k] : off;
    lv.off = m ? F.lv[k].off : lv.off;
    lv.mask = m ? F.lv[k].mask : lv.mask;
    lv.m1 = m ? F.lv[k].m1 : lv.m1;
    lv.m2 = m ? F.lv[k].m2 : lv.m2;
  }
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * 729; i += 256) acc[i] = 0.f;
  __syncthreads();
  const unsigned local = blockIdx.x - first, grp = local % groups, b = local / groups;
  const unsigned bx = b % nb, by = (b / nb) % nb, bz = b / (nb * nb);
  const unsigned long long cells = (unsigned long long)n * n * n;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned cl = tid + 256 * h, lx = cl & 7u, ly = (cl >> 3) & 7u, lz = cl >> 6;
    const unsigned x = bx * 8 + lx, y = by * 8 + ly, z = bz * 8 + lz;
    if (x >= n || y >= n || z >= n) continue;
    const unsigned long long cidx = x + (unsigned long long)n * (y + (unsigned long long)n * z);
    for (unsigned k = grp; k < copies; k += groups) {
      f32x4* p = reinterpret_cast<f32x4*>(F.c.base + off + ((unsigned long long)k * cells + cidx) * 16);
      const f32x4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
      const float v[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
      bool any = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) any = any || v[j] != 0.f;
      if (!any) continue;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      p[0] = zero;
      p[1] = zero;
      p[2] = zero;
      p[3] = zero;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned vi = ((lz + (c >> 2)) * 9 + (ly + ((c >> 1) & 1))) * 9 + (lx + (c & 1));
        if (v[2 * c] != 0.f) atomicAdd(&acc[2 * vi], v[2 * c]);
        if (v[2 * c + 1] != 0.f) atomicAdd(&acc[2 * vi + 1], v[2 * c + 1]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * 729; i += 256) {
    const float val = acc[i];
    if (val == 0.f) continue;
    const unsigned vtx = (unsigned)i >> 1, vx = vtx % 9, vy = (vtx / 9) % 9, vz = vtx / 81;
    const unsigned e = (((bx * 8 + vx) ^ ((by * 8 + vy) * lv.m1) ^ ((bz * 8 + vz) * lv.m2)) & lv.mask) + lv.off;
    cn_atomic_add(gtab + 2 * (size_t)e + (i & 1), val);
  }
}
inline void launch_cell_fold(const CellScatter& c, const GridDev& grid, float* gtab, hipStream_t stream) {
  if (!c.base || c.num_levels <= 0) return;
  const char* form = getenv("CN_CELL_FOLD");  // "records": the first form, one thread per record (A/B runs; always in the
  // deterministic test build: the block form sums in LDS with float atomics of four waves)
  if (!CN_DETERMINISTIC_SCATTER && (!form || strcmp(form, "records") != 0)) {
    CellFoldBlocksArgs B{};
    B.c = c;
    unsigned blocks = 0;
    for (int l = 0; l < c.num_levels; ++l) {
      B.first_block[l] = blocks;
      B.lv[l] = grid.level(l);
      B.nb[l] = (c.n[l] + 7) / 8;
      B.groups[l] = (c.copies[l] + 3) / 4;
      blocks += B.nb[l] * B.nb[l] * B.nb[l] * B.groups[l];
    }
    for (int l = c.num_levels; l <= CN_CELL_LEVELS; ++l) B.first_block[l] = blocks;
    for (int l = c.num_levels; l < CN_CELL_LEVELS; ++l) B.nb[l] = B.groups[l] = 1;
    hipLaunchKernelGGL(cell_scatter_fold_blocks_kernel, dim3(blocks), dim3(256), 0, stream, B, gtab);
    return;
  }
  CellFoldArgs F{};
  F.c = c;
  unsigned blocks = 0;
  for (int l = 0; l < c.num_levels; ++l) {
    F.first_block[l] = blocks;
    F.lv[l] = grid.level(l);
    const unsigned long long recs = (unsigned long long)c.n[l] * c.n[l] * c.n[l] * c.copies[l];
    blocks += (unsigned)((recs + 255) / 256);
  }
  for (int l = c.num_levels; l <= CN_CELL_LEVELS; ++l) F.first_block[l] = blocks;
  hipLaunchKernelGGL(cell_scatter_fold_kernel, dim3(blocks), dim3(256), 0, stream, F, gtab);
}

// fold the private copies into the gradient table and zero them again: 64 vertices of the dense n1^3 array per workgroup,
// the copies shared out over its 4 waves (each load is 64 consecutive float2 of one copy)
__global__ void __launch_bounds__(256) coarse_scatter_reduce_kernel(CoarseScatter c, Lvl lv, float* __restrict__ gtab) {
  __shared__ float2 part[4][64];
  const unsigned nv = c.n1 * c.n1 * c.n1;
  const unsigned j = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned v = blockIdx.x * 64u + j;
  float s0 = 0.f, s1 = 0.f;
  if (v < nv) {
    for (unsigned k = w; k < c.copies; k += 4) {
      float2* p = reinterpret_cast<float2*>(c.base) + (size_t)k * nv + v;
      const float2 t = *p;
      if (t.x != 0.f || t.y != 0.f) {
        s0 += t.x;
        s1 += t.y;
        *p = make_float2(0.f, 0.f);
      }
    }
  }
  part[w][j] = make_float2(s0, s1);
  __syncthreads();
  if (w != 0 || v >= nv) return;
  s0 = part[0][j].x + part[1][j].x + part[2][j].x + part[3][j].x;
  s1 = part[0][j].y + part[1][j].y + part[2][j].y + part[3][j].y;
  if (s0 == 0.f && s1 == 0.f) return;
  const unsigned x = v % c.n1, y = (v / c.n1) % c.n1, z = v / (c.n1 * c.n1);
  const unsigned e = ((x ^ (y * lv.m1) ^ (z * lv.m2)) & lv.mask) + lv.off;
  cn_atomic_add(gtab + 2 * (size_t)e, s0);
  cn_atomic_add(gtab + 2 * (size_t)e + 1, s1);
}
inline void launch_coarse_reduce(const CoarseScatter& c, const GridDev& grid, float* gtab, hipStream_t stream) {
  if (!c.base) return;
  const unsigned nv = c.n1 * c.n1 * c.n1;
  hipLaunchKernelGGL(coarse_scatter_reduce_kernel, dim3((nv + 63) / 64), dim3(256), 0, stream, c, grid.level(0), gtab);
}

// d(loss)/d(normalised position) -> d(loss)/d(world position): the transpose Jacobian of normalize_position
// (L-inf scene contraction then (c+2)/4, or the AABB normalisation), zero where the selector dropped the sample.
__device__ __forceinline__ void normalize_position_backward(const SceneDev& sc, float x, float y, float z, float sel,
                                                            float& gx, float& gy, float& gz) {
  if (sc.contraction) {
    gx *= 0.25f * sel;
    gy *= 0.25f * sel;
    gz *= 0.25f * sel;
    const float axv = fabsf(x), ayv = fabsf(y), azv = fabsf(z);
    const float m = fmaxf(axv, fmaxf(ayv, azv));
    if (m >= 1.f) {
      const float inv = 1.f / m;
      const float k = (2.f - inv) * inv;
      const float s = gx * x + gy * y + gz * z;
      const float coef = 2.f * inv * inv * (inv - 1.f) * s;  // d k / d m * <g, p>
      gx *= k;
      gy *= k;
      gz *= k;
      if (axv >= ayv && axv >= azv) gx += x < 0.f ? -coef : coef;
      else if (ayv >= azv) gy += y < 0.f ? -coef : coef;
      else gz += z < 0.f ? -coef : coef;
    }
  } else {
    gx *= sc.inv_extent[0] * sel;
    gy *= sc.inv_extent[1] * sel;
    gz *= sc.inv_extent[2] * sel;
  }
}

// d SH_deg4 / d (x, y, z) contracted with g[16] (the derivative of sh_deg4 in cn_common.hpp, term by term)
__device__ __forceinline__ void sh_deg4_backward(float x, float y, float z, const float* g, float& dx, float& dy,
                                                 float& dz) {
  const float xx = x * x, yy = y * y, zz = z * z;
  dx = 0.4886025119029199f * g[3] + 1.0925484305920792f * (y * g[4] + z * g[7]) + 1.0925484305920792f * x * g[8] +
       0.5900435899266435f * 6.f * x * y * g[9] + 2.890611442640554f * y * z * g[10] +
       0.4570457994644658f * (5.f * zz - 1.f) * g[13] + 1.445305721320277f * 2.f * x * z * g[14] +
       0.5900435899266435f * 3.f * (xx - yy) * g[15];
  dy = 0.4886025119029199f * g[1] + 1.0925484305920792f * (x * g[4] + z * g[5]) - 1.0925484305920792f * y * g[8] +
       0.5900435899266435f * 3.f * (xx - yy) * g[9] + 2.890611442640554f * x * z * g[10] +
       0.4570457994644658f * (5.f * zz - 1.f) * g[11] - 1.445305721320277f * 2.f * y * z * g[14] -
       0.5900435899266435f * 6.f * x * y * g[15];
  dz = 0.4886025119029199f * g[2] + 1.0925484305920792f * (y * g[5] + x * g[7]) + 0.9461746957575601f * 2.f * z * g[6] +
       2.890611442640554f * x * y * g[10] + 0.4570457994644658f * 10.f * z * (y * g[11] + x * g[13]) +
       0.3731763325901154f * (15.f * zz - 3.f) * g[12] + 1.445305721320277f * (xx - yy) * g[14];
}

struct FieldPtrs {
  const float* table;
  const float *w0, *b0, *w1, *b1;
  const float *ws0, *bs0, *ws1, *bs1, *wh, *bh;
  const float *wc0, *bc0, *wc1, *bc1, *wc2, *bc2;
  const float* emb;
};
struct FieldGrads {
  float* table;
  float *w0, *b0, *w1, *b1;
  float *ws0, *bs0, *ws1, *bs1, *wh, *bh;
  float *wc0, *bc0, *wc1, *bc1, *wc2, *bc2;
  float* emb;
};

struct FieldBwdArgs {
  FieldPtrs p;
  FieldGrads g;
  GridDev grid;  // geometry of p.table / g.table (fp32 tables)
  SceneDev scene;
  int sh_unit;
  int app_per_camera;
  const float* app_mean;  // [32] when not per-camera (may be null -> zeros)
  const float *origins, *directions, *starts, *ends;
  const int64_t* cam_idx;
  const float *d_density, *d_rgb, *d_sem;
  long long R;
  int S;
  float *d_pos, *d_dir;  // optional [R*S,3] outputs for the camera pose refinement (null: skipped)
  int debug_skip;  // profiling aid (env CN_DEBUG_SKIP): 1 hash atomics, 2 embedding atomics, 4 weight-gradient dots,
                   // 32 semantic branch, 64 forward gathers (matrix-core kernel), bits 8 + l: the scatter of level l
  CoarseScatter coarse;  // private copies for level 0's gradient (cn_grid.scatter_scratch of the gradient grid)
  CellScatter cells;     // cell-major records of the coarse levels (take precedence for the levels they cover)
};

// LDS rows (each LD floats)
constexpr int R_ENC = 0;            // 32
constexpr int R_H1 = R_ENC + 32;    // 64 (post ReLU)
constexpr int R_O16 = R_H1 + 64;    // 16
constexpr int R_S1 = R_O16 + 16;    // 64 (post ReLU)
constexpr int R_S2 = R_S1 + 64;     // 64
constexpr int R_CIN = R_S2 + 64;    // 63 (+1 pad row)
constexpr int R_C1 = R_CIN + 64;    // 64
constexpr int R_C2 = R_C1 + 64;     // 64
constexpr int R_DA = R_C2 + 64;     // 64 delta buffer A
constexpr int R_DB = R_DA + 64;     // 64 delta buffer B
// misc rows: 0-2 normalised position, 3 selector, 4 d(logit), 5 d(sem), 6-8 world position, 9-20 per-wave partial
// d(loss)/d(normalised position) (3 per wave)
constexpr int R_MISC = R_DB + 64;
constexpr int FIELD_ROWS = R_MISC + 21;

__global__ void __launch_bounds__(TB) field_backward_kernel(FieldBwdArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> weights come through s_load
  float* enc = lds + R_ENC * LD;
  float* h1 = lds + R_H1 * LD;
  float* o16 = lds + R_O16 * LD;
  float* s1 = lds + R_S1 * LD;
  float* s2 = lds + R_S2 * LD;
  float* cin = lds + R_CIN * LD;
  float* c1 = lds + R_C1 * LD;
  float* c2 = lds + R_C2 * LD;
  float* dA = lds + R_DA * LD;
  float* dB = lds + R_DB * LD;
  float* misc = lds + R_MISC * LD;

  WGrad<32, 64> gW0;
  WGrad<64, 16> gW1;
  WGrad<15, 64> gWs0;
  WGrad<64, 64> gWs1;
  WGrad<63, 64> gWc0;
  WGrad<64, 64> gWc1;
  WGrad<64, 3> gWc2;
  WGrad<64, 1> gWh;
  gW0.zero(); gW1.zero(); gWs0.zero(); gWs1.zero(); gWc0.zero(); gWc1.zero(); gWc2.zero(); gWh.zero();
  float gb0 = 0.f, gb1 = 0.f, gbs0 = 0.f, gbs1 = 0.f, gbh = 0.f, gbc0 = 0.f, gbc1 = 0.f, gbc2 = 0.f;

  const long long total = A.R * (long long)A.S;
  const long long ntiles = (total + TS - 1) / TS;
  // (one contiguous run of tiles per workgroup: train_field_mfma.hpp on batches sorted by camera and pixel)
  const long long tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
  const long long tile_end = ((long long)blockIdx.x + 1) * tiles_per_wg < ntiles ? ((long long)blockIdx.x + 1) * tiles_per_wg : ntiles;
  for (long long tile = blockIdx.x * tiles_per_wg; tile < tile_end; ++tile) {
    const long long i = tile * TS + lane;
    const bool valid = i < total;
    const long long ic = valid ? i : total - 1;
    const long long r = ic / A.S;
    // ---- per-sample inputs (wave 0 fills the shared rows) ------------------------------------------------------
    if (wave == 0) {
      const float mid = (A.starts[ic] + A.ends[ic]) / 2.f;
      float px = A.origins[3 * r] + A.directions[3 * r] * mid;
      float py = A.origins[3 * r + 1] + A.directions[3 * r + 1] * mid;
      float pz = A.origins[3 * r + 2] + A.directions[3 * r + 2] * mid;
      misc[6 * LD + lane] = px;
      misc[7 * LD + lane] = py;
      misc[8 * LD + lane] = pz;
      bool sel = normalize_position(A.scene, px, py, pz);
      misc[0 * LD + lane] = px;
      misc[1 * LD + lane] = py;
      misc[2 * LD + lane] = pz;
      misc[3 * LD + lane] = sel ? 1.f : 0.f;
      misc[5 * LD + lane] = valid ? A.d_sem[ic] : 0.f;
      // colour input: SH(16) | geo (filled after the base MLP) | appearance(32)
      float dx = A.directions[3 * r], dy = A.directions[3 * r + 1], dz = A.directions[3 * r + 2];
      if (!A.sh_unit) {
        dx = (dx + 1.f) / 2.f;
        dy = (dy + 1.f) / 2.f;
        dz = (dz + 1.f) / 2.f;
      }
      float sh[16];
      sh_deg4(dx, dy, dz, sh);
#pragma unroll
      for (int k = 0; k < 16; ++k) cin[k * LD + lane] = sh[k];
      const float* a = A.app_per_camera ? A.p.emb + A.cam_idx[r] * 32 : A.app_mean;
      for (int k = 0; k < 32; ++k) cin[(31 + k) * LD + lane] = a ? a[k] : 0.f;
    }
    __syncthreads();
    // ---- forward recompute -------------------------------------------------------------------------------------------
    {
      const float px = misc[0 * LD + lane], py = misc[1 * LD + lane], pz = misc[2 * LD + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int l = 4 * wave + q;
        float2 f = hash_level(A.p.table, A.grid.level(l), A.grid.pos_offset, px, py, pz);
        enc[(2 * l) * LD + lane] = f.x;
        enc[(2 * l + 1) * LD + lane] = f.y;
      }
    }
    __syncthreads();
    fwd_rows<32, 64, true>(A.p.w0, A.p.b0, enc, h1, wave, lane);
    __syncthreads();
    fwd_rows<64, 16, false>(A.p.w1, A.p.b1, h1, o16, wave, lane);
    __syncthreads();
    if (wave == 0) {
      // trunc_exp backward: g * exp(clamp(x, -15, 15)), times the selector; d_density is d loss / d (post-selector density)
      const float logit = o16[lane];
      const float dd = valid ? A.d_density[ic] : 0.f;
      misc[4 * LD + lane] = dd * misc[3 * LD + lane] * expf(fminf(fmaxf(logit, -15.f), 15.f));
    }
    for (int k = wave; k < 15; k += 4) cin[(16 + k) * LD + lane] = o16[(1 + k) * LD + lane];
    fwd_rows<15, 64, true>(A.p.ws0, A.p.bs0, o16 + LD, s1, wave, lane);  // geo = rows 1..15 of o16
    __syncthreads();
    fwd_rows<64, 64, false>(A.p.ws1, A.p.bs1, s1, s2, wave, lane);
    fwd_rows<63, 64, true>(A.p.wc0, A.p.bc0, cin, c1, wave, lane);
    __syncthreads();
    fwd_rows<64, 64, true>(A.p.wc1, A.p.bc1, c1, c2, wave, lane);
    __syncthreads();
    // ---- colour head: rgb = sigmoid(Wc2 c2 + bc2); delta_pre = d_rgb * rgb (1 - rgb) -> dA rows 0..2 ----------------
    if (wave < 3) {
      float acc = A.p.bc2[wave];
      for (int k = 0; k < 64; ++k) acc = fmaf(A.p.wc2[wave * 64 + k], c2[k * LD + lane], acc);
      const float s = 1.f / (1.f + expf(-acc));
      const float up = valid ? A.d_rgb[3 * ic + wave] : 0.f;
      dA[wave * LD + lane] = up * s * (1.f - s);
    }
    __syncthreads();
    if (!(A.debug_skip & 4)) gWc2.add(dA, c2, tid);
    bias_add<3>(gbc2, dA, tid);
    bwd_rows<64, 3>(A.p.wc2, dA, dB, c2, 0, 64, wave, lane);  // delta_c2 (ReLU-gated) -> dB
    __syncthreads();
    if (!(A.debug_skip & 4)) gWc1.add(dB, c1, tid);
    bias_add<64>(gbc1, dB, tid);
    bwd_rows<64, 64>(A.p.wc1, dB, dA, c1, 0, 64, wave, lane);  // delta_c1 -> dA
    __syncthreads();
    if (!(A.debug_skip & 4)) gWc0.add(dA, cin, tid);
    bias_add<64>(gbc0, dA, tid);
    // delta of the colour input: geo rows (16..30) feed the base MLP, appearance rows (31..62) the embedding
    // (rows 0..15, the SH inputs, only when the direction gradient is wanted)
    bwd_rows<63, 64>(A.p.wc0, dA, dB, nullptr, A.d_dir ? 0 : 16, 63, wave, lane);  // dB rows 16..62
    __syncthreads();
    if (A.app_per_camera && valid && !(A.debug_skip & 2)) {
      for (int k = wave; k < 32; k += 4) cn_atomic_add(A.g.emb + A.cam_idx[r] * 32 + k, dB[(31 + k) * LD + lane]);
    }
    if (A.d_dir && wave == 3 && valid) {
      float gsh[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) gsh[k] = dB[k * LD + lane];
      float dx = A.directions[3 * r], dy = A.directions[3 * r + 1], dz = A.directions[3 * r + 2];
      const float chain = A.sh_unit ? 1.f : 0.5f;
      if (!A.sh_unit) {
        dx = (dx + 1.f) / 2.f;
        dy = (dy + 1.f) / 2.f;
        dz = (dz + 1.f) / 2.f;
      }
      float gx, gy, gz;
      sh_deg4_backward(dx, dy, dz, gsh, gx, gy, gz);
      A.d_dir[3 * i] = gx * chain;
      A.d_dir[3 * i + 1] = gy * chain;
      A.d_dir[3 * i + 2] = gz * chain;
    }
    // delta_o16 -> dA' : row 0 = density logit, rows 1..15 = geo (from the colour branch only: semantics sees detached geo)
    // (dA is still needed by nobody: gWc0 has consumed it)
    __syncthreads();
    if (wave == 0) dA[lane] = misc[4 * LD + lane];
    for (int k = wave; k < 15; k += 4) dA[(1 + k) * LD + lane] = dB[(16 + k) * LD + lane];
    __syncthreads();
    if (!(A.debug_skip & 4)) gW1.add(dA, h1, tid);
    bias_add<16>(gb1, dA, tid);
    bwd_rows<64, 16>(A.p.w1, dA, dB, h1, 0, 64, wave, lane);  // delta_h1 -> dB
    __syncthreads();
    if (!(A.debug_skip & 4)) gW0.add(dB, enc, tid);
    bias_add<64>(gb0, dB, tid);
    bwd_rows<32, 64>(A.p.w0, dB, dA, nullptr, 0, 32, wave, lane);  // delta_enc -> dA rows 0..31
    __syncthreads();
    if (!(A.debug_skip & 1)) {
      const float px = misc[0 * LD + lane], py = misc[1 * LD + lane], pz = misc[2 * LD + lane];
      float gpx = 0.f, gpy = 0.f, gpz = 0.f;
      if (A.d_pos) {  // kernel-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int l = 4 * wave + q;
          hash_level_backward<true>(A.g.table, A.p.table, A.grid.level(l), A.grid.pos_offset, px, py, pz,
                                    valid ? dA[(2 * l) * LD + lane] : 0.f, valid ? dA[(2 * l + 1) * LD + lane] : 0.f,
                                    lane, gpx, gpy, gpz);
        }
        misc[(9 + 3 * wave) * LD + lane] = gpx;
        misc[(10 + 3 * wave) * LD + lane] = gpy;
        misc[(11 + 3 * wave) * LD + lane] = gpz;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int l = 4 * wave + q;
          hash_level_backward<false>(A.g.table, A.p.table, A.grid.level(l), A.grid.pos_offset, px, py, pz,
                                     valid ? dA[(2 * l) * LD + lane] : 0.f, valid ? dA[(2 * l + 1) * LD + lane] : 0.f,
                                     lane, gpx, gpy, gpz);
        }
      }
    }
    __syncthreads();
    if (A.d_pos && wave == 3 && valid) {
      float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        gx += misc[(9 + 3 * w) * LD + lane];
        gy += misc[(10 + 3 * w) * LD + lane];
        gz += misc[(11 + 3 * w) * LD + lane];
      }
      normalize_position_backward(A.scene, misc[6 * LD + lane], misc[7 * LD + lane], misc[8 * LD + lane],
                                  misc[3 * LD + lane], gx, gy, gz);
      A.d_pos[3 * i] = gx;
      A.d_pos[3 * i + 1] = gy;
      A.d_pos[3 * i + 2] = gz;
    }
    // ---- semantic branch: sem = Wh s2 + bh; gradients stop at the (detached) geo features ------------------------------
    // delta_sem (1 row) is misc row 5
    if (!(A.debug_skip & 4)) gWh.add(misc + 5 * LD, s2, tid);
    bias_add<1>(gbh, misc + 5 * LD, tid);
    bwd_rows<64, 1>(A.p.wh, misc + 5 * LD, dB, nullptr, 0, 64, wave, lane);  // delta_s2 -> dB
    __syncthreads();
    if (!(A.debug_skip & 4)) gWs1.add(dB, s1, tid);
    bias_add<64>(gbs1, dB, tid);
    bwd_rows<64, 64>(A.p.ws1, dB, dA, s1, 0, 64, wave, lane);  // delta_s1 -> dA
    __syncthreads();
    if (!(A.debug_skip & 4)) gWs0.add(dA, o16 + LD, tid);
    bias_add<64>(gbs0, dA, tid);
    __syncthreads();
  }
  gW0.flush(A.g.w0, tid); gW1.flush(A.g.w1, tid); gWs0.flush(A.g.ws0, tid); gWs1.flush(A.g.ws1, tid);
  gWc0.flush(A.g.wc0, tid); gWc1.flush(A.g.wc1, tid); gWc2.flush(A.g.wc2, tid); gWh.flush(A.g.wh, tid);
  if (tid < 64) {
    cn_atomic_add(A.g.b0 + tid, gb0);
    cn_atomic_add(A.g.bs0 + tid, gbs0);
    cn_atomic_add(A.g.bs1 + tid, gbs1);
    cn_atomic_add(A.g.bc0 + tid, gbc0);
    cn_atomic_add(A.g.bc1 + tid, gbc1);
  }
  if (tid < 16) cn_atomic_add(A.g.b1 + tid, gb1);
  if (tid < 3) cn_atomic_add(A.g.bc2 + tid, gbc2);
  if (tid < 1) cn_atomic_add(A.g.bh + tid, gbh);
}

}  // namespace cn
#include "train_field_mfma.hpp"
namespace cn {

}  // namespace cn
#include "train_field_general.hpp"
namespace cn {

// ------------------------------------------------------------------------------------------------------------------------
// proposal network backward
// ------------------------------------------------------------------------------------------------------------------------
struct PropBwdArgs {
  const float* table;
  const float *w0, *b0, *w1, *b1;
  float *g_table, *g_w0, *g_b0, *g_w1, *g_b1;
  GridDev grid;
  SceneDev scene;
  const float *origins, *directions, *starts, *ends, *d_density;
  float* d_pos;  // optional [R*S,3]
  long long R;
  int S;
  int debug_skip;  // CN_DEBUG_SKIP: 8 hash atomics, 16 weight-gradient dots
  CoarseScatter coarse;
  CellScatter cells;  // cell-major records of the coarse levels (takes precedence over `coarse` for the levels it covers)
};

template <int L>
__global__ void __launch_bounds__(TB, 4) proposal_backward_kernel(PropBwdArgs A) {
  constexpr int K = 2 * L, H = 16;
  __shared__ float lds[(K + H + H + 1 + 4 + 3 + 12) * LD + 4 * 64 * 17];
  float* enc = lds;                  // [K]
  float* hid = enc + K * LD;         // [H] post ReLU
  float* dh = hid + H * LD;          // [H] delta hidden
  float* dout = dh + H * LD;         // [1] delta logit
  float* misc = dout + LD;           // normalised pos(3) sel(1) world pos(3) per-wave d(pos)(12)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> weights come through s_load
  float* tb = misc + 19 * LD + wave * (64 * 17);  // this wave's transpose buffer of hash_level_backward_cells
  WGrad<K, H> gW0;
  WGrad<H, 1> gW1;
  gW0.zero();
  gW1.zero();
  float gb0 = 0.f, gb1 = 0.f;
  const long long total = A.R * (long long)A.S;
  const long long ntiles = (total + TS - 1) / TS;
  // (one contiguous run of tiles per workgroup: train_field_mfma.hpp on batches sorted by camera and pixel)
  const long long tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
  const long long tile_end = ((long long)blockIdx.x + 1) * tiles_per_wg < ntiles ? ((long long)blockIdx.x + 1) * tiles_per_wg : ntiles;
  for (long long tile = blockIdx.x * tiles_per_wg; tile < tile_end; ++tile) {
    const long long i = tile * TS + lane;
    const bool valid = i < total;
    const long long ic = valid ? i : total - 1;
    const long long r = ic / A.S;
    float sel_f = 0.f;
    if (wave == 0) {
      const float mid = (A.starts[ic] + A.ends[ic]) / 2.f;
      float px = A.origins[3 * r] + A.directions[3 * r] * mid;
      float py = A.origins[3 * r + 1] + A.directions[3 * r + 1] * mid;
      float pz = A.origins[3 * r + 2] + A.directions[3 * r + 2] * mid;
      misc[4 * LD + lane] = px;
      misc[5 * LD + lane] = py;
      misc[6 * LD + lane] = pz;
      bool sel = normalize_position(A.scene, px, py, pz);
      sel_f = sel ? 1.f : 0.f;
      misc[0 * LD + lane] = px;
      misc[1 * LD + lane] = py;
      misc[2 * LD + lane] = pz;
      misc[3 * LD + lane] = sel_f;
    }
    __syncthreads();
    // (levels shared out as in the scatter below, so that the Jacobian of a level's features with respect to the position --
    //  hash_level_jac: the position gradient without a second gather -- stays in the registers of the wave that needs it)
    v2f_t jx[2], jy[2], jz[2];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int l = round == 0 ? L - 1 - wave : L - 8 + wave;
      jx[round] = jy[round] = jz[round] = v2f_t{0.f, 0.f};
      if (l < 0) continue;
      float2 f = hash_level_jac(A.table, A.grid.level(l), A.grid.pos_offset, misc[lane], misc[LD + lane],
                                misc[2 * LD + lane], jx[round], jy[round], jz[round]);
      enc[(2 * l) * LD + lane] = f.x;
      enc[(2 * l + 1) * LD + lane] = f.y;
    }
    __syncthreads();
    fwd_rows<K, H, true>(A.w0, A.b0, enc, hid, wave, lane);
    __syncthreads();
    if (wave == 0) {
      float logit = A.b1[0];
#pragma unroll
      for (int k = 0; k < H; ++k) logit = fmaf(A.w1[k], hid[k * LD + lane], logit);
      const float up = valid ? A.d_density[ic] : 0.f;
      dout[lane] = up * misc[3 * LD + lane] * expf(fminf(fmaxf(logit, -15.f), 15.f));
    }
    __syncthreads();
    if (!(A.debug_skip & 16)) gW1.add(dout, hid, tid);
    bias_add<1>(gb1, dout, tid);
    bwd_rows<H, 1>(A.w1, dout, dh, hid, 0, H, wave, lane);
    __syncthreads();
    if (!(A.debug_skip & 16)) gW0.add(dh, enc, tid);
    bias_add<H>(gb0, dh, tid);
    // delta_enc[k] = sum_n W0[n][k] dh[n] -> straight into the table gradient
    float gpx = 0.f, gpy = 0.f, gpz = 0.f;
    // the finest level costs most (one request per x-edge, no runs to merge) and the coarsest least: the waves take the
    // levels from the fine end, the second round from the other side (L = 5: {4}, {3}, {2}, {1, 0})
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int l = round == 0 ? L - 1 - wave : L - 8 + wave;
      if (l < 0) continue;
      float g0 = 0.f, g1 = 0.f;
#pragma unroll
      for (int n = 0; n < H; ++n) {
        const float d = dh[n * LD + lane];
        g0 = fmaf(A.w0[n * K + 2 * l], d, g0);
        g1 = fmaf(A.w0[n * K + 2 * l + 1], d, g1);
      }
      if ((A.debug_skip & 8) || ((A.debug_skip >> (8 + l)) & 1)) continue;  // bits 8..14: skip the scatter of level l (profiling)
      g0 = valid ? g0 : 0.f;
      g1 = valid ? g1 : 0.f;
      gpx += g0 * jx[round].x + g1 * jx[round].y;
      gpy += g0 * jy[round].x + g1 * jy[round].y;
      gpz += g0 * jz[round].x + g1 * jz[round].y;
      float ux = 0.f, uy = 0.f, uz = 0.f;  // (unused: the <false> forms do not touch them)
      if (l < A.cells.num_levels) {
        const unsigned nl = A.cells.n[l];
        float* rec = A.cells.base + A.cells.offset[l] +
                     (size_t)(blockIdx.x % A.cells.copies[l]) * ((size_t)nl * nl * nl * 16);
        hash_level_backward_cells<false>(rec, nl, tb, A.g_table, A.table, A.grid.level(l), A.grid.pos_offset, misc[lane],
                                         misc[LD + lane], misc[2 * LD + lane], g0, g1, lane, ux, uy, uz);
      } else if (l == 0 && A.coarse.base) {
        float* mine = A.coarse.base + (size_t)(blockIdx.x % A.coarse.copies) * (2u * A.coarse.n1 * A.coarse.n1 * A.coarse.n1);
        hash_level_backward_private<false>(mine, A.coarse.n1, A.g_table, A.table, A.grid.level(0), A.grid.pos_offset,
                                           misc[lane], misc[LD + lane], misc[2 * LD + lane], g0, g1, lane, ux, uy, uz);
      } else
        hash_level_backward<false>(A.g_table, A.table, A.grid.level(l), A.grid.pos_offset, misc[lane],
                                   misc[LD + lane], misc[2 * LD + lane], g0, g1, lane, ux, uy, uz);
    }
    if (A.d_pos) {
      misc[(7 + 3 * wave) * LD + lane] = gpx;
      misc[(8 + 3 * wave) * LD + lane] = gpy;
      misc[(9 + 3 * wave) * LD + lane] = gpz;
      __syncthreads();
      if (wave == 0 && valid) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          gx += misc[(7 + 3 * w) * LD + lane];
          gy += misc[(8 + 3 * w) * LD + lane];
          gz += misc[(9 + 3 * w) * LD + lane];
        }
        normalize_position_backward(A.scene, misc[4 * LD + lane], misc[5 * LD + lane], misc[6 * LD + lane],
                                    misc[3 * LD + lane], gx, gy, gz);
        A.d_pos[3 * i] = gx;
        A.d_pos[3 * i + 1] = gy;
        A.d_pos[3 * i + 2] = gz;
      }
    }
    __syncthreads();
  }
  gW0.flush(A.g_w0, tid);
  gW1.flush(A.g_w1, tid);
  if (tid < H) cn_atomic_add(A.g_b0 + tid, gb0);
  if (tid < 1) cn_atomic_add(A.g_b1 + tid, gb1);
}

}  // namespace cn
#include "train_proposal_wave.hpp"
namespace cn {

static double cell_scatter_ratio(bool proposal, double dflt);
int validate_field(const cn_field_params& p);  // field_simple.hip
int validate_grid(const cn_grid& g, const char* name);

static bool is_default_field_shape(const cn_field_params& p) {
  return p.grid.num_levels == 16 && p.geo_feat_dim == 15 && p.app_dim == 32 && p.base.num_layers == 2 &&
         p.base.dims[0] == 32 && p.base.dims[1] == 64 && p.base.dims[2] == 16 && p.semantics.num_layers == 2 &&
         p.semantics.dims[0] == 15 && p.semantics.dims[1] == 64 && p.semantics.dims[2] == 64 &&
         p.color.num_layers == 3 && p.color.dims[0] == 63 && p.color.dims[1] == 64 && p.color.dims[2] == 64 &&
         p.color.dims[3] == 3;
}

}  // namespace cn

extern "C" int cn_field_backward(const cn_field_params* params, const cn_field_params* grads, const cn_scene* scene,
                                 int32_t app_mode, int32_t sh_unit_dir, const float* app_mean, const float* origins,
                                 const float* directions, const int64_t* camera_indices, const float* starts,
                                 const float* ends, const float* d_density, const float* d_rgb, const float* d_semantics,
                                 int64_t num_rays, int32_t num_samples, float* d_positions, float* d_directions,
                                 cn_stream_t stream) {
  return cn_field_backward_mp(params, grads, scene, app_mode, sh_unit_dir, app_mean, origins, directions, camera_indices, starts,
                              ends, d_density, d_rgb, d_semantics, num_rays, num_samples, d_positions, d_directions,
                              CN_MATRIX_FP32, stream);
}

extern "C" int cn_field_backward_mp(const cn_field_params* params, const cn_field_params* grads, const cn_scene* scene,
                                    int32_t app_mode, int32_t sh_unit_dir, const float* app_mean, const float* origins,
                                    const float* directions, const int64_t* camera_indices, const float* starts,
                                    const float* ends, const float* d_density, const float* d_rgb, const float* d_semantics,
                                    int64_t num_rays, int32_t num_samples, float* d_positions, float* d_directions,
                                    int32_t matrix_precision, cn_stream_t stream) {
  CN_REQUIRE(params && grads && scene && origins && directions && starts && ends && d_density && d_rgb && d_semantics,
             CN_ERR_INVALID, "cn_field_backward: null argument");
  CN_REQUIRE(matrix_precision == CN_MATRIX_FP32 || matrix_precision == CN_MATRIX_SPLIT_BF16 || matrix_precision == CN_MATRIX_F16,
             CN_ERR_INVALID, "cn_field_backward: matrix_precision %d", matrix_precision);
  CN_REQUIRE(app_mode != CN_APP_PER_CAMERA || camera_indices, CN_ERR_INVALID, "Camera indices are not provided.");
  CN_REQUIRE(app_mode != CN_APP_MEAN || app_mean, CN_ERR_INVALID, "cn_field_backward: app_mean required for CN_APP_MEAN");
  int rc = cn::validate_field(*params);
  if (rc) return rc;
  if ((rc = cn::validate_field(*grads))) return rc;
  CN_REQUIRE(cn::is_default_field_shape(*params) && cn::is_default_field_shape(*grads), CN_ERR_UNSUPPORTED,
             "cn_field_backward is built for the default fruit_nerf_method field shape");
  if ((rc = cn::check_grad_grid(params->grid, grads->grid, "cn_field_backward"))) return rc;
  if (num_rays <= 0) return CN_OK;
  cn::FieldBwdArgs A{};
  auto fill = [](auto& dst, const cn_field_params& s) {
    dst.w0 = (decltype(dst.w0))s.base.weight[0];
    dst.b0 = (decltype(dst.b0))s.base.bias[0];
    dst.w1 = (decltype(dst.w1))s.base.weight[1];
    dst.b1 = (decltype(dst.b1))s.base.bias[1];
    dst.ws0 = (decltype(dst.ws0))s.semantics.weight[0];
    dst.bs0 = (decltype(dst.bs0))s.semantics.bias[0];
    dst.ws1 = (decltype(dst.ws1))s.semantics.weight[1];
    dst.bs1 = (decltype(dst.bs1))s.semantics.bias[1];
    dst.wh = (decltype(dst.wh))s.sem_head_weight;
    dst.bh = (decltype(dst.bh))s.sem_head_bias;
    dst.wc0 = (decltype(dst.wc0))s.color.weight[0];
    dst.bc0 = (decltype(dst.bc0))s.color.bias[0];
    dst.wc1 = (decltype(dst.wc1))s.color.weight[1];
    dst.bc1 = (decltype(dst.bc1))s.color.bias[1];
    dst.wc2 = (decltype(dst.wc2))s.color.weight[2];
    dst.bc2 = (decltype(dst.bc2))s.color.bias[2];
    dst.emb = (decltype(dst.emb))s.appearance;
    dst.table = (decltype(dst.table))s.grid.table;
  };
  fill(A.p, *params);
  fill(A.g, *grads);
  A.grid = cn::make_grid_dev(params->grid);
  A.scene = cn::make_scene_dev(*scene);
  A.sh_unit = sh_unit_dir;
  A.app_per_camera = app_mode == CN_APP_PER_CAMERA;
  A.app_mean = app_mode == CN_APP_MEAN ? app_mean : nullptr;
  A.origins = origins;
  A.directions = directions;
  A.starts = starts;
  A.ends = ends;
  A.cam_idx = camera_indices;
  A.d_density = d_density;
  A.d_rgb = d_rgb;
  A.d_sem = d_semantics;
  A.d_pos = d_positions;
  A.d_dir = d_directions;
  A.R = num_rays;
  A.S = num_samples;
  {
    const char* dbg = getenv("CN_DEBUG_SKIP");
    A.debug_skip = dbg ? atoi(dbg) : 0;
  }
  // default: the matrix-core kernel; CN_FIELD_BACKWARD_IMPL=scalar selects the first (scalar-FMA) implementation,
  // kept as an independent device implementation for cross-checks
  const char* impl_env = getenv("CN_FIELD_BACKWARD_IMPL");  // read per call: one process can compare both
  const bool use_scalar = impl_env && std::strcmp(impl_env, "scalar") == 0;
  static cn::PerDevice<int> attrs;  // one-time kernel attributes, per device
  rc = attrs.get(
      [](int, int&) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::field_backward_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((size_t)cn::FIELD_ROWS * cn::LD * sizeof(float)));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::mf::field_backward_mfma_kernel<0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cn::mf::LDS_BYTES);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(cn::mf::field_backward_mfma_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)cn::mf::LDS_BYTES);
      },
      nullptr, "cn_field_backward");
  if (rc) return rc;
  const long long nsamp = num_rays * (long long)num_samples;
  if (use_scalar) {
    size_t lds = (size_t)cn::FIELD_ROWS * cn::LD * sizeof(float);
    long long ntiles = (nsamp + cn::TS - 1) / cn::TS;
    hipLaunchKernelGGL(cn::field_backward_kernel, dim3(cn::grid_for(ntiles, 1, 256)), dim3(cn::TB), lds,
                       cn::as_stream(stream), A);
  } else {
    long long ntiles = (nsamp + cn::mf::TSM - 1) / cn::mf::TSM;
    A.coarse = cn::make_coarse_scatter(grads->grid);
    {  // cell-major records for the levels with at most CELL_RATIO_FIELD cells per sample (CN_CELL_SCATTER=<ratio>, 0: off)
      const double ratio = cn::cell_scatter_ratio(false, cn::CELL_RATIO_FIELD);
      if (ratio != 0.0) A.cells = cn::make_cell_scatter(grads->grid, (unsigned long long)(nsamp * ratio), (unsigned long long)nsamp);
      if (A.cells.num_levels > 0) A.coarse.base = nullptr;  // level 0 is cell-major then
    }
    // (split-bf16 keeps ~fp32 products in the forward; its gradient is the exact-fp32 kernel's)
    if (matrix_precision == CN_MATRIX_F16)
      hipLaunchKernelGGL(cn::mf::field_backward_mfma_kernel<1>, dim3(cn::grid_for(ntiles, 1, 256)), dim3(cn::mf::NT),
                         cn::mf::LDS_BYTES, cn::as_stream(stream), A);
    else
      hipLaunchKernelGGL(cn::mf::field_backward_mfma_kernel<0>, dim3(cn::grid_for(ntiles, 1, 256)), dim3(cn::mf::NT),
                         cn::mf::LDS_BYTES, cn::as_stream(stream), A);
    CN_DET_FLUSH(cn::as_stream(stream));  // (deterministic test build: the scratch records are floats again before the folds)
    cn::launch_coarse_reduce(A.coarse, A.grid, A.g.table, cn::as_stream(stream));
    cn::launch_cell_fold(A.cells, A.grid, A.g.table, cn::as_stream(stream));
  }
  if (int rc2 = cn::check_launch("cn_field_backward")) return rc2;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

namespace cn {
// cells-per-sample ratio up to which a level's gradient goes through cell-major records:
// CN_CELL_SCATTER sets it for every backward kernel (0 = off), CN_CELL_SCATTER_PROP for the proposal networks alone.
static double cell_scatter_ratio(bool proposal, double dflt) {
  const char* cs = getenv("CN_CELL_SCATTER");
  const char* cp = proposal ? getenv("CN_CELL_SCATTER_PROP") : nullptr;
  if (cp) return atof(cp);
  return cs ? atof(cs) : dflt;
}
}  // namespace cn

extern "C" int cn_proposal_backward(const cn_density_params* params, const cn_density_params* grads,
                                    const cn_scene* scene, const float* origins, const float* directions,
                                    const float* starts, const float* ends, const float* d_density, int64_t num_rays,
                                    int32_t num_samples, float* d_positions, cn_stream_t stream) {
  CN_REQUIRE(params && grads && scene && origins && directions && starts && ends && d_density, CN_ERR_INVALID,
             "cn_proposal_backward: null argument");
  int rc = cn::check_grad_grid(params->grid, grads->grid, "cn_proposal_backward");
  if (rc) return rc;
  const int L = params->grid.num_levels;
  bool ok = (L == 5 || L == 7) && params->mlp.num_layers == 2 && params->mlp.dims[0] == 2 * L &&
            params->mlp.dims[1] == 16 && params->mlp.dims[2] == 1 && grads->grid.num_levels == L &&
            grads->grid.log2_table_size == params->grid.log2_table_size;
  CN_REQUIRE(ok, CN_ERR_UNSUPPORTED, "cn_proposal_backward: proposal net must be {5|7 levels, 2L->16->1}");
  CN_REQUIRE(grads->grid.table && grads->mlp.weight[0] && grads->mlp.bias[0] && grads->mlp.weight[1] &&
                 grads->mlp.bias[1],
             CN_ERR_INVALID, "cn_proposal_backward: null gradient buffer");
  if (num_rays <= 0) return CN_OK;
  cn::PropBwdArgs A{};
  A.table = static_cast<const float*>(params->grid.table);
  A.w0 = params->mlp.weight[0];
  A.b0 = params->mlp.bias[0];
  A.w1 = params->mlp.weight[1];
  A.b1 = params->mlp.bias[1];
  A.g_table = static_cast<float*>(const_cast<void*>(grads->grid.table));
  A.g_w0 = const_cast<float*>(grads->mlp.weight[0]);
  A.g_b0 = const_cast<float*>(grads->mlp.bias[0]);
  A.g_w1 = const_cast<float*>(grads->mlp.weight[1]);
  A.g_b1 = const_cast<float*>(grads->mlp.bias[1]);
  A.grid = cn::make_grid_dev(params->grid);
  A.scene = cn::make_scene_dev(*scene);
  A.origins = origins;
  A.directions = directions;
  A.starts = starts;
  A.ends = ends;
  A.d_density = d_density;
  A.d_pos = d_positions;
  A.R = num_rays;
  A.S = num_samples;
  {
    const char* dbg = getenv("CN_DEBUG_SKIP");
    A.debug_skip = dbg ? atoi(dbg) : 0;
  }
  A.coarse = cn::make_coarse_scatter(grads->grid);
  // cell-major records for the levels with at most CELL_RATIO_PROPOSAL cells per sample (runs of a ray's samples merge there);
  // CN_CELL_SCATTER_PROP / CN_CELL_SCATTER = 0 keeps every level on the table path
  {
    const double ratio = cn::cell_scatter_ratio(true, cn::CELL_RATIO_PROPOSAL);
    const unsigned long long nsamp = (unsigned long long)num_rays * (unsigned long long)num_samples;
    if (ratio != 0.0) A.cells = cn::make_cell_scatter(grads->grid, (unsigned long long)(nsamp * ratio), nsamp);
    if (A.cells.num_levels > 0 && A.coarse.base) A.coarse.base = nullptr;  // level 0 is cell-major then
  }
  long long ntiles = (num_rays * (long long)num_samples + cn::TS - 1) / cn::TS;
  // one wave per tile (train_proposal_wave.hpp); CN_PROP_BWD=tile: the first form, four waves per tile (A/B runs, cross-check)
  const char* form = getenv("CN_PROP_BWD");
  if (form && strcmp(form, "tile") == 0) {
    dim3 grid(cn::grid_for(ntiles, 1, 1024));
    if (L == 5)
      hipLaunchKernelGGL(cn::proposal_backward_kernel<5>, grid, dim3(cn::TB), 0, cn::as_stream(stream), A);
    else
      hipLaunchKernelGGL(cn::proposal_backward_kernel<7>, grid, dim3(cn::TB), 0, cn::as_stream(stream), A);
  } else {
    dim3 grid(cn::grid_for((ntiles + 3) / 4, 1, 1024));
    if (L == 5)
      hipLaunchKernelGGL(cn::pw::proposal_backward_wave_kernel<5>, grid, dim3(256), 0, cn::as_stream(stream), A);
    else
      hipLaunchKernelGGL(cn::pw::proposal_backward_wave_kernel<7>, grid, dim3(256), 0, cn::as_stream(stream), A);
  }
  CN_DET_FLUSH(cn::as_stream(stream));
  cn::launch_coarse_reduce(A.coarse, A.grid, A.g_table, cn::as_stream(stream));
  cn::launch_cell_fold(A.cells, A.grid, A.g_table, cn::as_stream(stream));
  if (int rc2 = cn::check_launch("cn_proposal_backward")) return rc2;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" size_t cn_grid_scatter_scratch_bytes(const cn_grid* grid) {
  if (!grid || grid->num_levels < 1) return 0;
  const size_t head = cn::coarse_scratch_bytes(*grid);
  return head ? head + cn::cell_scratch_layout(*grid, nullptr) : 0;
}

// The same, sized for batches of at most `max_samples` samples per backward call: a level is kept cell-major only when it
// has at most (CN_CELL_SCATTER / CN_CELL_SCATTER_PROP; defaults CELL_RATIO_FIELD / CELL_RATIO_PROPOSAL, at most
// CELL_RATIO_MAX) x samples cells, so the records of levels with more than CELL_RATIO_MAX x max_samples cells would never be
// touched.  max_samples <= 0: every level up to CELL_MAX_CELLS cells (= cn_grid_scatter_scratch_bytes).
extern "C" size_t cn_grid_scatter_scratch_bytes_for(const cn_grid* grid, int64_t max_samples) {
  if (!grid || grid->num_levels < 1) return 0;
  const size_t head = cn::coarse_scratch_bytes(*grid);
  if (!head) return 0;
  cn::CellScatter c{};
  const size_t all = cn::cell_scratch_layout(*grid, &c);
  if (max_samples <= 0) return head + all;
  size_t bytes = 0;
  for (int l = 0; l < c.num_levels; ++l) {
    const unsigned long long cells = (unsigned long long)c.n[l] * c.n[l] * c.n[l];
    if ((double)cells > cn::CELL_RATIO_MAX * (double)max_samples) break;
    bytes = (size_t)(c.offset[l] + c.copies[l] * cells * 16ull) * sizeof(float);
  }
  return head + bytes;
}

// ---- shape-generic field backward ------------------------------------------------------------------------------------------
namespace cn {
static bool general_family_ok(const cn_field_params& p) {
  auto w128 = [](const cn_mlp& m) {
    for (int l = 0; l <= m.num_layers; ++l)
      if (m.dims[l] > 128) return false;
    return true;
  };
  return p.base.num_layers == 2 && (p.semantics.num_layers == 2 || p.semantics.num_layers == 3) &&
         p.color.num_layers == 3 && w128(p.base) && w128(p.semantics) && w128(p.color) && p.geo_feat_dim <= 30 &&
         2 * p.grid.num_levels <= 32 && 16 + p.geo_feat_dim + p.app_dim <= 128;
}
// floats of one workgroup's scratch slice: every parameter tensor starts on a 64-byte line (a 16-float segment of a weight
// row that straddles two lines costs two L2 requests each way: 5 240 write requests per tile instead of 3 212, measured)
static int general_pad16(int n) { return (n + 15) & ~15; }
static int general_param_count(const cn_field_params& p) {
  auto mlp = [](const cn_mlp& m) {
    int n = 0;
    for (int l = 0; l < m.num_layers; ++l) n += general_pad16(m.dims[l] * m.dims[l + 1]) + general_pad16(m.dims[l + 1]);
    return n;
  };
  return mlp(p.base) + mlp(p.semantics) + mlp(p.color) + general_pad16(p.semantics.dims[p.semantics.num_layers]) + 16;
}
static int general_blocks() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus;
}
}  // namespace cn

extern "C" size_t cn_field_backward_general_workspace_bytes(const cn_field_params* params) {
  if (!params) return 0;
  return (size_t)cn::general_blocks() * ((cn::general_param_count(*params) + 3) / 4 * 4) * sizeof(float);
}

extern "C" int cn_field_backward_general(const cn_field_params* params, const cn_field_params* grads,
                                         const cn_scene* scene, int32_t app_mode, int32_t sh_unit_dir,
                                         const float* app_mean, const float* origins, const float* directions,
                                         const int64_t* camera_indices, const float* starts, const float* ends,
                                         const float* d_density, const float* d_rgb, const float* d_semantics,
                                         int64_t num_rays, int32_t num_samples, float* d_positions,
                                         float* d_directions, void* workspace, size_t workspace_bytes,
                                         cn_stream_t stream) {
  CN_REQUIRE(params && grads && scene && origins && directions && starts && ends && d_density && d_rgb && d_semantics,
             CN_ERR_INVALID, "cn_field_backward_general: null argument");
  CN_REQUIRE(app_mode != CN_APP_PER_CAMERA || camera_indices, CN_ERR_INVALID, "Camera indices are not provided.");
  CN_REQUIRE(app_mode != CN_APP_MEAN || app_mean, CN_ERR_INVALID, "cn_field_backward_general: app_mean required");
  int rc = cn::validate_field(*params);
  if (rc) return rc;
  if ((rc = cn::validate_field(*grads))) return rc;
  CN_REQUIRE(cn::general_family_ok(*params), CN_ERR_UNSUPPORTED,
             "cn_field_backward_general: base 2 layers, semantics 2-3 layers, colour 3 layers, widths <= 128");
  if ((rc = cn::check_grad_grid(params->grid, grads->grid, "cn_field_backward_general"))) return rc;
  if (num_rays <= 0) return CN_OK;
  const int nblk = cn::general_blocks();
  const int ppb = (cn::general_param_count(*params) + 3) / 4 * 4;
  CN_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * ppb * sizeof(float), CN_ERR_WORKSPACE,
             "cn_field_backward_general: workspace %zu B < %zu B", workspace_bytes, (size_t)nblk * ppb * sizeof(float));
  hipStream_t s = cn::as_stream(stream);
  cn::gb::GenArgs A{};
  int off = 0;
  struct Target { float* g; int off, n; };
  Target targets[24];
  int nt = 0;
  auto fill = [&](cn::gb::GenLayer* dst, const cn_mlp& m, const cn_mlp& gm) {
    for (int l = 0; l < m.num_layers; ++l) {
      dst[l].W = m.weight[l];
      dst[l].b = m.bias[l];
      dst[l].K = m.dims[l];
      dst[l].N = m.dims[l + 1];
      dst[l].off_w = off;
      targets[nt++] = {const_cast<float*>(gm.weight[l]), off, m.dims[l] * m.dims[l + 1]};
      off += cn::general_pad16(m.dims[l] * m.dims[l + 1]);
      dst[l].off_b = off;
      targets[nt++] = {const_cast<float*>(gm.bias[l]), off, m.dims[l + 1]};
      off += cn::general_pad16(m.dims[l + 1]);
    }
  };
  fill(A.base, params->base, grads->base);
  fill(A.sem, params->semantics, grads->semantics);
  fill(A.col, params->color, grads->color);
  A.ns = params->semantics.num_layers;
  const int ht = params->semantics.dims[A.ns];
  A.wh = params->sem_head_weight;
  A.off_wh = off;
  targets[nt++] = {const_cast<float*>(grads->sem_head_weight), off, ht};
  off += cn::general_pad16(ht);
  A.off_bh = off;
  targets[nt++] = {const_cast<float*>(grads->sem_head_bias), off, 1};
  off += 16;
  CN_REQUIRE(off == cn::general_param_count(*params), CN_ERR_WORKSPACE, "cn_field_backward_general: scratch layout %d != %d", off,
             cn::general_param_count(*params));
  for (int i = 0; i < nt; ++i) CN_REQUIRE(targets[i].g, CN_ERR_INVALID, "cn_field_backward_general: null gradient buffer");
  A.params_per_block = ppb;
  A.scratch = static_cast<float*>(workspace);
  auto p16 = [](int n) { return (n + 15) & ~15; };
  int rows = 0, wmax = 16;
  auto take = [&](int n) { int r = rows; rows += n; return r; };
  const int enc = 2 * params->grid.num_levels, cin = 16 + params->geo_feat_dim + params->app_dim;
  for (const cn_mlp* m : {&params->base, &params->semantics, &params->color})
    for (int l = 0; l <= m->num_layers; ++l) wmax = std::max(wmax, p16(m->dims[l]));
  A.r_enc = take(p16(enc));
  A.r_h1 = take(p16(params->base.dims[1]));
  A.r_g = take(48);
  for (int l = 0; l < A.ns; ++l) A.r_s[l] = take(p16(params->semantics.dims[l + 1]));
  A.r_cin = take(p16(cin));
  A.r_c1 = take(p16(params->color.dims[1]));
  A.r_c2 = take(p16(params->color.dims[2]));
  A.r_rgb = take(16);
  A.r_da = take(std::max(wmax, 48));  // also the 16 x 3 position-gradient partials
  A.r_db = take(wmax);
  A.r_dcin = take(p16(cin));
  A.r_dg = take(32);
  A.r_drgb = take(16);
  A.r_dsem = take(16);
  A.rows = rows;
  const size_t lds = ((size_t)rows * cn::gb::LDG + cn::LVL_REC_FLOATS) * sizeof(float);
  CN_REQUIRE(lds <= 160 * 1024, CN_ERR_UNSUPPORTED,
             "cn_field_backward_general: the activations of one 32-sample tile need %zu B of LDS (max 163840)", lds);
  A.table = static_cast<const float*>(params->grid.table);
  A.g_table = static_cast<float*>(const_cast<void*>(grads->grid.table));
  A.emb = params->appearance;
  A.g_emb = const_cast<float*>(grads->appearance);
  A.app_mean = app_mode == CN_APP_MEAN ? app_mean : nullptr;
  A.grid = cn::make_grid_dev(params->grid);
  A.num_levels = params->grid.num_levels;
  A.geo = params->geo_feat_dim;
  A.app_dim = params->app_dim;
  A.app_per_camera = app_mode == CN_APP_PER_CAMERA;
  A.sh_unit = sh_unit_dir;
  A.scene = cn::make_scene_dev(*scene);
  A.origins = origins;
  A.directions = directions;
  A.starts = starts;
  A.ends = ends;
  A.cam_idx = camera_indices;
  A.d_density = d_density;
  A.d_rgb = d_rgb;
  A.d_sem = d_semantics;
  A.d_pos = d_positions;
  A.d_dir = d_directions;
  A.R = num_rays;
  A.S = num_samples;
  CN_REQUIRE(A.g_table && (!A.app_per_camera || A.g_emb), CN_ERR_INVALID, "cn_field_backward_general: null gradient buffer");
  // (the LDS need depends on the field shape, so the attribute is set per call: cheap, and correct on every device)
  CN_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(cn::gb::field_backward_general_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
             CN_ERR_LAUNCH, "cn_field_backward_general: hipFuncSetAttribute(MaxDynamicSharedMemorySize, %zu) failed", lds);
  CN_REQUIRE(hipMemsetAsync(workspace, 0, (size_t)nblk * ppb * sizeof(float), s) == hipSuccess, CN_ERR_LAUNCH,
             "cn_field_backward_general: hipMemsetAsync failed");
  const long long ntiles = (num_rays * (long long)num_samples + cn::gb::TSG - 1) / cn::gb::TSG;
  const int grid = (int)std::min<long long>(ntiles, nblk);
  A.coarse = cn::make_coarse_scatter(grads->grid);
  {  // cell-major records (as in cn_field_backward) when the four LDS buffers that carry the hand-over hold two waves each
    const double ratio = cn::cell_scatter_ratio(false, cn::CELL_RATIO_FIELD);
    const int small = std::min(std::min(p16(params->color.dims[1]), p16(params->color.dims[2])), p16(cin));
    if (ratio != 0.0 && small * cn::gb::LDG >= 2 * 64 * 17 && A.num_levels <= 16)
      A.cells = cn::make_cell_scatter(grads->grid, (unsigned long long)(num_rays * (double)num_samples * ratio),
                                      (unsigned long long)num_rays * (unsigned long long)num_samples);
    if (A.cells.num_levels > 0) A.coarse.base = nullptr;
  }
  {
    const char* e = getenv("CN_DEBUG_SKIP");  // profiling aid (results are wrong by construction)
    A.debug_skip = e ? atoi(e) : 0;
  }
  hipLaunchKernelGGL(cn::gb::field_backward_general_kernel, dim3(grid), dim3(cn::gb::NTG), lds, s, A);
  CN_DET_FLUSH(s);
  cn::launch_coarse_reduce(A.coarse, A.grid, A.g_table, s);
  cn::launch_cell_fold(A.cells, A.grid, A.g_table, s);
  rc = cn::check_launch("cn_field_backward_general");
  if (rc) return rc;
  CN_DET_FLUSH(s);
  cn::gb::ReduceTargets T;
  T.count = nt;
  for (int i = 0; i < nt; ++i) {
    T.g[i] = targets[i].g;
    T.off[i] = targets[i].off;
    T.n[i] = targets[i].n;
  }
  hipLaunchKernelGGL(cn::gb::field_backward_reduce_all_kernel, dim3((ppb + 31) / 32), dim3(256), 0, s, A.scratch, grid, ppb, T);
  return cn::check_launch("cn_field_backward_general reduce");
}

// Shared host/device helpers for libcropnerf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "../../include/cropnerf_hip.h"

#define CN_WAVE 64

namespace cn {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CN_ERR_LAUNCH;
  }
  return CN_OK;
}

#define CN_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) {                  \
      cn::set_error(__VA_ARGS__);   \
      return (code);                \
    }                               \
  } while (0)

inline hipStream_t as_stream(cn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// One-time, PER-DEVICE kernel setup (hipFuncSetAttribute(MaxDynamicSharedMemorySize), occupancy queries): a process that
// uses several devices through the C ABI gets the attributes set on each of them, and a failed attribute call surfaces
// as CN_ERR_LAUNCH with a message instead of a later opaque launch error.  init(device, value) returns a hipError_t.
constexpr int CN_MAX_DEVICES = 64;
template <typename T>
struct PerDevice {
  std::once_flag once[CN_MAX_DEVICES];
  hipError_t err[CN_MAX_DEVICES] = {};
  T value[CN_MAX_DEVICES] = {};
  template <typename F>
  int get(F&& init, const T** out, const char* who) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= CN_MAX_DEVICES) {
      set_error("%s: no current HIP device (%s)", who, hipGetErrorString(e));
      return CN_ERR_LAUNCH;
    }
    std::call_once(once[dev], [&] { err[dev] = init(dev, value[dev]); });
    if (err[dev] != hipSuccess) {
      set_error("%s: one-time kernel setup failed on device %d: %s", who, dev, hipGetErrorString(err[dev]));
      return CN_ERR_LAUNCH;
    }
    if (out) *out = &value[dev];
    return CN_OK;
  }
};

inline unsigned grid_for(long long n, int block, long long cap = 1 << 20) {
  long long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------

#define CN_P1 2654435761u
#define CN_P2 805459861u

// One level of a grid as the kernels see it.  entry = off + ((hx ^ hy ^ hz) & mask) with hx = ix (+1), hy = iy * m1 (+ m1),
// hz = iz * m2 (+ m2): for a hashed level m1, m2 are the hash primes and mask = T - 1; for a dense level of the tcnn layout
// m1 = 2^b, m2 = 2^2b, mask = 2^3b - 1 (disjoint bit fields: the xor IS the sum) -- one expression, no per-level branch.
struct Lvl {
  unsigned off, mask, m1, m2;
  float scale;
};

struct GridDev {  // by-value kernel argument
  const void* table;
  int num_levels;
  int half;          // entries are half2 (CN_TABLE_F16) instead of float2
  float pos_offset;  // cell = floor(x * scale + pos_offset): 0 (nerfstudio torch HashEncoding) or 0.5 (tcnn)
  unsigned off[CN_MAX_LEVELS], mask[CN_MAX_LEVELS], m1[CN_MAX_LEVELS], m2[CN_MAX_LEVELS];
  float scale[CN_MAX_LEVELS];
  __host__ __device__ __forceinline__ Lvl level(int l) const {  // l wave-uniform (scalar loads from the kernarg segment)
    Lvl v;
    v.off = off[l];
    v.mask = mask[l];
    v.m1 = m1[l];
    v.m2 = m2[l];
    v.scale = scale[l];
    return v;
  }
  __host__ __device__ __forceinline__ const float* table_f32() const { return static_cast<const float*>(table); }
};

inline GridDev make_grid_dev(const cn_grid& g) {
  GridDev d;
  d.table = g.table;
  d.num_levels = g.num_levels;
  d.half = g.table_dtype == CN_TABLE_F16;
  d.pos_offset = g.layout == CN_GRID_TCNN ? 0.5f : 0.f;
  const unsigned T = 1u << g.log2_table_size;
  for (int i = 0; i < CN_MAX_LEVELS; ++i) {
    const bool live = i < g.num_levels;
    d.scale[i] = live ? g.scalings[i] : 0.f;
    const int b = (live && g.layout == CN_GRID_TCNN) ? g.level_bits[i] : 0;
    d.off[i] = !live ? 0u : (g.layout == CN_GRID_TCNN ? g.level_offset[i] : (unsigned)i * T);
    d.m1[i] = b ? 1u << b : CN_P1;
    d.m2[i] = b ? 1u << (2 * b) : CN_P2;
    d.mask[i] = b ? (1u << (3 * b)) - 1u : T - 1u;
  }
  return d;
}

// Host-side validation shared by every entry point that takes a cn_grid.  need_f32: kernels that write gradients or
// were not built for half tables.
inline int check_grid(const cn_grid& g, bool need_f32, const char* who) {
  CN_REQUIRE(g.table, CN_ERR_INVALID, "%s: null hash table", who);
  CN_REQUIRE(g.num_levels >= 1 && g.num_levels <= CN_MAX_LEVELS, CN_ERR_UNSUPPORTED, "%s: %d grid levels", who,
             g.num_levels);
  CN_REQUIRE(g.layout == CN_GRID_TORCH || g.layout == CN_GRID_TCNN, CN_ERR_INVALID, "%s: grid layout %d", who, g.layout);
  CN_REQUIRE(g.table_dtype == CN_TABLE_F32 || g.table_dtype == CN_TABLE_F16, CN_ERR_INVALID, "%s: table dtype %d", who,
             g.table_dtype);
  CN_REQUIRE(!need_f32 || g.table_dtype == CN_TABLE_F32, CN_ERR_UNSUPPORTED,
             "%s: needs an fp32 hash table (half tables are inference-only)", who);
  CN_REQUIRE(g.log2_table_size >= 1 && g.log2_table_size <= 24, CN_ERR_UNSUPPORTED, "%s: log2 table size %d", who,
             g.log2_table_size);
  // every gather address is a 32-bit byte offset from the table base
  unsigned long long entries = (unsigned long long)g.num_levels << g.log2_table_size;
  if (g.layout == CN_GRID_TCNN) {
    entries = 0;
    for (int l = 0; l < g.num_levels; ++l) {
      const int b = g.level_bits[l];
      CN_REQUIRE(b <= 9, CN_ERR_UNSUPPORTED, "%s: dense level %d with %d bits per axis", who, l, b);
      // the x-pair gathers (hash_level_xpair_issue / hash_level_pk_issue) read entries e and e ^ 1 as ONE aligned pair
      CN_REQUIRE(g.level_offset[l] % 2 == 0, CN_ERR_INVALID, "%s: level %d starts at the odd entry %u", who, l,
                 g.level_offset[l]);
      const unsigned long long end = (unsigned long long)g.level_offset[l] + (b ? 1ull << (3 * b) : 1ull << g.log2_table_size);
      if (end > entries) entries = end;
    }
  }
  CN_REQUIRE(entries * 8ull <= (1ull << 31), CN_ERR_UNSUPPORTED, "%s: hash table larger than 2 GiB", who);
  // an aligned entry pair is one dwordx4 load of a float table, one dwordx2 load of a half table
  const uintptr_t align = g.table_dtype == CN_TABLE_F16 ? 7u : 15u;
  CN_REQUIRE((reinterpret_cast<uintptr_t>(g.table) & align) == 0, CN_ERR_INVALID,
             "%s: the hash table must be %u-byte aligned", who, (unsigned)align + 1u);
  return CN_OK;
}

// Backward entry points: the gradient table mirrors the parameter table entry for entry, both fp32.
inline int check_grad_grid(const cn_grid& p, const cn_grid& g, const char* who) {
  int rc = check_grid(p, true, who);
  if (rc) return rc;
  if ((rc = check_grid(g, true, who))) return rc;
  bool same = p.layout == g.layout && p.num_levels == g.num_levels && p.log2_table_size == g.log2_table_size;
  if (same && p.layout == CN_GRID_TCNN)
    for (int l = 0; l < p.num_levels; ++l)
      same = same && p.level_offset[l] == g.level_offset[l] && p.level_bits[l] == g.level_bits[l];
  CN_REQUIRE(same, CN_ERR_INVALID, "%s: the gradient table's geometry differs from the parameter table's", who);
  return CN_OK;
}

struct SceneDev {
  float lo[3];
  float inv_extent[3];  // 1 / (hi - lo), host-computed
  int contraction;
};

inline SceneDev make_scene_dev(const cn_scene& s) {
  SceneDev d;
  for (int i = 0; i < 3; ++i) {
    d.lo[i] = s.aabb[i];
    d.inv_extent[i] = 1.f / (s.aabb[3 + i] - s.aabb[i]);
  }
  d.contraction = s.contraction;
  return d;
}

// Normalised position + selector (fruit_field.py:171-180).  Returns selector; p is zeroed when deselected.
// Select-only (no divergent branches): the contraction scale uses one v_rcp_f32 and the AABB normalisation a
// host-side reciprocal, i.e. positions may differ from the oracle's divisions by an ulp or two -- the encoding is
// continuous in the position, so this stays inside the fp32 parity tolerance.
__device__ __forceinline__ bool normalize_position(const SceneDev& sc, float& x, float& y, float& z) {
  if (sc.contraction) {  // wave-uniform branch
    float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    float inv = __builtin_amdgcn_rcpf(m);
    float k = m < 1.f ? 1.f : (2.f - inv) * inv;
    x = (x * k + 2.f) * 0.25f;
    y = (y * k + 2.f) * 0.25f;
    z = (z * k + 2.f) * 0.25f;
  } else {
    x = (x - sc.lo[0]) * sc.inv_extent[0];
    y = (y - sc.lo[1]) * sc.inv_extent[1];
    z = (z - sc.lo[2]) * sc.inv_extent[2];
  }
  bool sel = (x > 0.f) && (x < 1.f) && (y > 0.f) && (y < 1.f) && (z > 0.f) && (z < 1.f);
  x = sel ? x : 0.f;
  y = sel ? y : 0.f;
  z = sel ? z : 0.f;
  return sel;
}

// One level of the hash grid: 8 corner gathers of two features + trilinear blend in the order of
// HashEncoding.pytorch_fwd (x toward the upper corner, then y, then z).  tcnn's kernel_grid sums the same eight
// weight * value products corner by corner; the results differ in fp32 rounding only.
// `table` is the wave-uniform base of the whole table and lv.off the (possibly per-lane) level offset in entries: every
// gather address is base + 32-bit byte offset (global_load saddr+voffset form), so a gather costs one address VGPR and
// no 64-bit adds.  entries * 8 B <= 2^31 is checked on the host (check_grid).
template <bool HALF = false>
__device__ __forceinline__ float2 hash_gather(const void* __restrict__ table, unsigned entry) {
  if constexpr (HALF) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = *reinterpret_cast<const h2*>(reinterpret_cast<const char*>(table) + (size_t)(entry << 2));
    return make_float2((float)v.x, (float)v.y);
  } else {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(table) + (size_t)(entry << 3));
  }
}

// cell coordinates / in-cell offsets / the per-axis index terms of the 8 corners (upper corner = lower + 1: identical to
// the ceil corner of the torch fallback unless the coordinate is integral, where that corner's weight is 0)
struct Cell {
  float ox, oy, oz;
  unsigned hx0, hx1, hy0, hy1, hz0, hz1;
};
// OFFSET = false: pos_offset is known to be 0 (torch layout in the kernels specialised for it): plain products, which
// hipcc contracts with the subtraction below exactly as the round-1 kernels did.
template <bool OFFSET = true>
__device__ __forceinline__ Cell hash_cell(const Lvl& lv, float pos_offset, float px, float py, float pz) {
  const float sx = OFFSET ? fmaf(px, lv.scale, pos_offset) : px * lv.scale;
  const float sy = OFFSET ? fmaf(py, lv.scale, pos_offset) : py * lv.scale;
  const float sz = OFFSET ? fmaf(pz, lv.scale, pos_offset) : pz * lv.scale;
  const float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  Cell c;
  c.ox = sx - fx;
  c.oy = sy - fy;
  c.oz = sz - fz;
  const unsigned ix = (unsigned)(int)fx, iy = (unsigned)(int)fy, iz = (unsigned)(int)fz;
  c.hx0 = ix;
  c.hx1 = ix + 1u;
  // 24-bit multiplies (v_mul_u32_u24, full rate; a 32-bit v_mul_lo_u32 is a quarter-rate instruction): cell indices are far
  // below 2^24, a dense level's strides too (exact product), and of a hash prime only the low 24 bits matter -- the index is
  // masked to at most 24 bits (check_grid) and the low 24 bits of a product depend on the low 24 bits of its factors only.
  c.hy0 = __umul24(iy, lv.m1);
  c.hy1 = c.hy0 + lv.m1;
  c.hz0 = __umul24(iz, lv.m2);
  c.hz1 = c.hz0 + lv.m2;
  return c;
}

template <bool HALF = false>
__device__ __forceinline__ float2 hash_level(const void* __restrict__ table, const Lvl& lv, float pos_offset, float px,
                                             float py, float pz) {
  const Cell k = hash_cell(lv, pos_offset, px, py, pz);
  const float ox = k.ox, oy = k.oy, oz = k.oz;
  float2 ccc = hash_gather<HALF>(table, ((k.hx1 ^ k.hy1 ^ k.hz1) & lv.mask) + lv.off);  // f_0
  float2 cfc = hash_gather<HALF>(table, ((k.hx1 ^ k.hy0 ^ k.hz1) & lv.mask) + lv.off);  // f_1
  float2 ffc = hash_gather<HALF>(table, ((k.hx0 ^ k.hy0 ^ k.hz1) & lv.mask) + lv.off);  // f_2
  float2 fcc = hash_gather<HALF>(table, ((k.hx0 ^ k.hy1 ^ k.hz1) & lv.mask) + lv.off);  // f_3
  float2 ccf = hash_gather<HALF>(table, ((k.hx1 ^ k.hy1 ^ k.hz0) & lv.mask) + lv.off);  // f_4
  float2 cff = hash_gather<HALF>(table, ((k.hx1 ^ k.hy0 ^ k.hz0) & lv.mask) + lv.off);  // f_5
  float2 fff = hash_gather<HALF>(table, ((k.hx0 ^ k.hy0 ^ k.hz0) & lv.mask) + lv.off);  // f_6
  float2 fcf = hash_gather<HALF>(table, ((k.hx0 ^ k.hy1 ^ k.hz0) & lv.mask) + lv.off);  // f_7
  float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
  // The two features of a corner sit in one register pair: blending them as 2-vectors with scalar weights maps onto
  // v_pk_mul_f32 / v_pk_fma_f32 with the weight broadcast by op_sel, with no register shuffling (written per component,
  // hipcc's SLP vectoriser packs ACROSS corners instead and pays ~18 v_mov per level and sample for it).
  typedef float v2f __attribute__((ext_vector_type(2)));
  auto V = [](float2 t) {
    v2f v;
    v.x = t.x;
    v.y = t.y;
    return v;
  };
  const v2f f03 = V(ccc) * ox + V(fcc) * mx, f12 = V(cfc) * ox + V(ffc) * mx;
  const v2f f56 = V(cff) * ox + V(fff) * mx, f47 = V(ccf) * ox + V(fcf) * mx;
  const v2f a = f03 * oy + f12 * my, b = f47 * oy + f56 * my;
  const v2f rv = a * oz + b * mz;
  float2 r;
  r.x = rv.x;
  r.y = rv.y;
  return r;
}

// hash_level that also returns the Jacobian of the two features with respect to the NORMALISED position (d f / d x etc., the
// level's scale included): the trilinear blend is linear in each in-cell offset, so the derivatives are differences of the
// partial blends the interpolation forms anyway.  The training backward takes the position gradient of a (sample, level) as
// g0 * J.x + g1 * J.y from these six numbers instead of gathering the eight corners a second time once g is known.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 hash_level_jac(const void* __restrict__ table, const Lvl& lv, float pos_offset, float px, float py,
                                                 float pz, v2f_t& jx, v2f_t& jy, v2f_t& jz) {
  const Cell k = hash_cell(lv, pos_offset, px, py, pz);
  const float ox = k.ox, oy = k.oy, oz = k.oz;
  float2 ccc = hash_gather<false>(table, ((k.hx1 ^ k.hy1 ^ k.hz1) & lv.mask) + lv.off);
  float2 cfc = hash_gather<false>(table, ((k.hx1 ^ k.hy0 ^ k.hz1) & lv.mask) + lv.off);
  float2 ffc = hash_gather<false>(table, ((k.hx0 ^ k.hy0 ^ k.hz1) & lv.mask) + lv.off);
  float2 fcc = hash_gather<false>(table, ((k.hx0 ^ k.hy1 ^ k.hz1) & lv.mask) + lv.off);
  float2 ccf = hash_gather<false>(table, ((k.hx1 ^ k.hy1 ^ k.hz0) & lv.mask) + lv.off);
  float2 cff = hash_gather<false>(table, ((k.hx1 ^ k.hy0 ^ k.hz0) & lv.mask) + lv.off);
  float2 fff = hash_gather<false>(table, ((k.hx0 ^ k.hy0 ^ k.hz0) & lv.mask) + lv.off);
  float2 fcf = hash_gather<false>(table, ((k.hx0 ^ k.hy1 ^ k.hz0) & lv.mask) + lv.off);
  const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
  auto V = [](float2 t) {
    v2f_t v;
    v.x = t.x;
    v.y = t.y;
    return v;
  };
  const v2f_t f03 = V(ccc) * ox + V(fcc) * mx, f12 = V(cfc) * ox + V(ffc) * mx;
  const v2f_t f56 = V(cff) * ox + V(fff) * mx, f47 = V(ccf) * ox + V(fcf) * mx;
  const v2f_t a = f03 * oy + f12 * my, b = f47 * oy + f56 * my;
  const v2f_t rv = a * oz + b * mz;
  // x: the four x-edges' differences blended over y and z; y: the (y = 1) minus the (y = 0) partial blends over z; z: a - b
  const v2f_t d03 = V(ccc) - V(fcc), d12 = V(cfc) - V(ffc), d56 = V(cff) - V(fff), d47 = V(ccf) - V(fcf);
  jx = ((d03 * oy + d12 * my) * oz + (d47 * oy + d56 * my) * mz) * lv.scale;
  jy = ((f03 - f12) * oz + (f47 - f56) * mz) * lv.scale;
  jz = (a - b) * lv.scale;
  float2 r;
  r.x = rv.x;
  r.y = rv.y;
  return r;
}

// hash_level with x-pair gathers (round 3).  A CU's L1 looks up one cache line per clock for per-lane-addressed loads
// (tools/gather_rate_microbench.hip), and the gather-heavy kernels run at that rate; the two x-corners of a (y, z) row are
// entries e and e ^ 1 -- one aligned pair -- whenever the cell's x index is even (hashed levels: index = x ^ y*P1 ^ z*P2;
// dense levels of the tcnn layout: x in the low bits).  So each row is ONE load of the aligned pair that holds the lower
// corner (16 bytes of float2 entries, 8 bytes of half2 entries: one lookup either way), and lanes with an odd x index fetch
// their upper corner with a second load under the execution mask: 6 lookups per level instead of 8.  Same values, same
// blend as hash_level (results may differ in the last bit where hipcc contracts the multiply-adds differently).  The two
// results are pinned with an empty asm: the conditional load is control flow, and without it hipcc sinks the blends of
// all levels behind the last level's loads.
// Two halves, so that a caller can put other work -- the next level's loads -- between them: `issue` sends the four pair loads
// AND, for lanes with an odd x index, the four single loads of the upper corners back to back (nothing between them reads a loaded
// value: one round trip to the cache, not two); `blend` waits and interpolates.
template <bool HALF>
struct XpLoads {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typename std::conditional<HALF, h4, f4>::type pr[4];   // aligned pairs, rows (y, z) = ff, cf, fc, cc
  typename std::conditional<HALF, h2, float2>::type up[4];  // upper-x corners of lanes with an odd x index
  unsigned bits;  // bit r: the lower corner is the SECOND entry of pair r; bit 4: odd x index
  float ox, oy, oz;
};
template <bool HALF = false>
__device__ __forceinline__ XpLoads<HALF> hash_level_xpair_issue(const void* __restrict__ table, const Lvl& lv, float pos_offset,
                                                                float px, float py, float pz) {
  const Cell k = hash_cell(lv, pos_offset, px, py, pz);
  const char* base = reinterpret_cast<const char*>(table);
  const unsigned pair_mask = lv.mask & ~1u;
  const unsigned hyz[4] = {k.hy0 ^ k.hz0, k.hy1 ^ k.hz0, k.hy0 ^ k.hz1, k.hy1 ^ k.hz1};
  const bool odd = (k.hx0 & 1u) != 0u;
  XpLoads<HALF> L;
  L.bits = odd ? 16u : 0u;
  L.ox = k.ox;
  L.oy = k.oy;
  L.oz = k.oz;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned x = k.hx0 ^ hyz[r];
    L.bits |= (x & 1u) << r;
    const unsigned e = (x & pair_mask) + lv.off;  // first entry of the aligned pair (level offsets are even)
    if constexpr (HALF) L.pr[r] = *reinterpret_cast<const typename XpLoads<HALF>::h4*>(base + (size_t)(e << 2));
    else L.pr[r] = *reinterpret_cast<const typename XpLoads<HALF>::f4*>(base + (size_t)(e << 3));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if constexpr (HALF) L.up[r] = typename XpLoads<HALF>::h2{(_Float16)0.f, (_Float16)0.f};
    else L.up[r] = make_float2(0.f, 0.f);
  }
  if (odd) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned e = ((k.hx1 ^ hyz[r]) & lv.mask) + lv.off;
      if constexpr (HALF) L.up[r] = *reinterpret_cast<const typename XpLoads<HALF>::h2*>(base + (size_t)(e << 2));
      else L.up[r] = *reinterpret_cast<const float2*>(base + (size_t)(e << 3));
    }
  }
  return L;
}
#ifndef CN_XPAIR_WEIGHT_SWAP
#define CN_XPAIR_WEIGHT_SWAP 1
#endif
template <bool HALF = false>
__device__ __forceinline__ float2 hash_level_xpair_blend(const XpLoads<HALF>& L) {
  const bool odd = (L.bits & 16u) != 0u;
#if CN_XPAIR_WEIGHT_SWAP
  // Which entry of an aligned pair is the lower-x corner depends on the row's hash parity; instead of SELECTING the values (six
  // conditional moves per row: lo / hi of two features, and the separately loaded upper corner of odd cells) the x WEIGHTS are
  // selected -- row value = a wa + b wb + u ox with (wa, wb) = (mx, ox') or (ox', mx), ox' = 0 for odd cells (their upper corner
  // is u; u is zero for even cells): two conditional moves per row and one more packed multiply-add.
  {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float ox = L.ox, oy = L.oy, oz = L.oz;
    const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
    const float oxe = odd ? 0.f : ox;
    v2f row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool second = ((L.bits >> r) & 1u) != 0u;
      const float wa = second ? oxe : mx, wb = second ? mx : oxe;
      v2f a, b, u;
      a.x = (float)L.pr[r].x;
      a.y = (float)L.pr[r].y;
      b.x = (float)L.pr[r].z;
      b.y = (float)L.pr[r].w;
      u.x = (float)L.up[r].x;
      u.y = (float)L.up[r].y;
      row[r] = u * ox + (b * wb + a * wa);
    }
    const v2f a2 = row[3] * oy + row[2] * my, b2 = row[1] * oy + row[0] * my;
    const v2f rv = a2 * oz + b2 * mz;
    float2 r;
    r.x = rv.x;
    r.y = rv.y;
    asm volatile("" : "+v"(r.x), "+v"(r.y));
    return r;
  }
#endif
  float2 lo[4], hi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool second = ((L.bits >> r) & 1u) != 0u;
    const float2 a = make_float2((float)L.pr[r].x, (float)L.pr[r].y), b = make_float2((float)L.pr[r].z, (float)L.pr[r].w);
    const float2 u = make_float2((float)L.up[r].x, (float)L.up[r].y);
    lo[r] = second ? b : a;
    hi[r] = odd ? u : (second ? a : b);  // the other entry of the pair is the upper-x corner when x is even
  }
  const float ox = L.ox, oy = L.oy, oz = L.oz;
  const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
  typedef float v2f __attribute__((ext_vector_type(2)));
  auto V = [](float2 t) {
    v2f v;
    v.x = t.x;
    v.y = t.y;
    return v;
  };
  const v2f f03 = V(hi[3]) * ox + V(lo[3]) * mx, f12 = V(hi[2]) * ox + V(lo[2]) * mx;
  const v2f f56 = V(hi[0]) * ox + V(lo[0]) * mx, f47 = V(hi[1]) * ox + V(lo[1]) * mx;
  const v2f a2 = f03 * oy + f12 * my, b2 = f47 * oy + f56 * my;
  const v2f rv = a2 * oz + b2 * mz;
  float2 r;
  r.x = rv.x;
  r.y = rv.y;
  asm volatile("" : "+v"(r.x), "+v"(r.y));
  return r;
}
template <bool HALF = false>
__device__ __forceinline__ float2 hash_level_xpair(const void* __restrict__ table, const Lvl& lv, float pos_offset, float px,
                                                   float py, float pz) {
  return hash_level_xpair_blend<HALF>(hash_level_xpair_issue<HALF>(table, lv, pos_offset, px, py, pz));
}

// Private accumulation of the coarsest level's gradient (cn_grid.scatter_scratch): `copies` dense [n1^3][2] arrays,
// vertex (x, y, z) at x + n1 * (y + n1 * z); a workgroup adds to copy blockIdx.x % copies.  base == nullptr: off.
struct CoarseScatter {
  float* base;
  unsigned n1, copies;
};
constexpr unsigned COARSE_COPIES = 64, COARSE_MIN_COPIES = 8, COARSE_MAX_N1 = 40;
// vertices per axis that level 0 can address for positions in [0, 1]: floor(scale + offset) is the largest cell index
inline unsigned coarse_n1(const cn_grid& g) {
  const float off = g.layout == CN_GRID_TCNN ? 0.5f : 0.f;
  return (unsigned)floorf(g.scalings[0] + off) + 2u;
}
inline CoarseScatter make_coarse_scatter(const cn_grid& grads_grid) {
  CoarseScatter c{nullptr, 0u, 0u};
  if (!grads_grid.scatter_scratch || grads_grid.num_levels < 1) return c;
  const unsigned n1 = coarse_n1(grads_grid);
  if (n1 > COARSE_MAX_N1) return c;
  const size_t per_copy = (size_t)n1 * n1 * n1 * 2 * sizeof(float);
  size_t copies = grads_grid.scatter_scratch_bytes / per_copy;
  if (copies > COARSE_COPIES) copies = COARSE_COPIES;
  if (copies < COARSE_MIN_COPIES) return c;
  c.base = static_cast<float*>(grads_grid.scatter_scratch);
  c.n1 = n1;
  c.copies = (unsigned)copies;
  return c;
}

// Cell-major gradient records of the coarse levels (cn_grid.scatter_scratch, behind the level-0 vertex copies): level l < num_levels
// keeps copies[l] arrays of n[l]^3 records of 16 floats -- the 8 corners x 2 features of ONE cell, corner c = a + 2 b + 4 d at
// floats 2c, 2c + 1 -- so that a sample adds its whole cell in ONE 64-byte request (the hash table takes 4.5: one per x-edge),
// and consecutive samples of a ray in the same cell merge into one.  A fold kernel adds the touched records to the table and
// zeroes them.  Worth it where samples outnumber cells: the launch picks the levels by batch size.
constexpr int CN_CELL_LEVELS = 10;
// A level goes through cell-major records when it has at most (ratio x samples of the call) cells.  Since the fold works by
// blocks (cell_scatter_fold_blocks_kernel: ~0.4 requests per cell and a streaming pass over the records) a level pays as long as
// one request per run of samples plus that pass is cheaper than 4.5 requests per sample: measured optimum (tools/train_probe.py,
// 4 096 / 65 536 rays) at ~2-3 cells per sample for the field (48 samples per ray: few samples share a cell at the
// fine levels; 3.03 -- 9.5e6 cells, 610 MB of records at 65 536 rays -- already costs 0.6 ms) and 3-8 for the proposal networks
// (256 / 96 samples per ray: runs merge).  CELL_RATIO_MAX bounds what cn_grid_scatter_scratch_bytes_for sizes the scratch for.
constexpr double CELL_RATIO_FIELD = 2.85, CELL_RATIO_PROPOSAL = 6.0, CELL_RATIO_MAX = 8.0;
constexpr unsigned long long CELL_MAX_CELLS = 17500000ull;  // 259^3 fits
struct CellScatter {
  float* base;  // nullptr: off
  int num_levels;
  unsigned n[CN_CELL_LEVELS];
  unsigned copies[CN_CELL_LEVELS];
  unsigned long long offset[CN_CELL_LEVELS];  // in floats from base
};
inline unsigned cell_n(const cn_grid& g, int l) {  // cells per axis that positions in [0, 1] can fall into
  const float off = g.layout == CN_GRID_TCNN ? 0.5f : 0.f;
  return (unsigned)floorf(g.scalings[l] + off) + 1u;
}
inline unsigned cell_copies(unsigned long long ncells) { return ncells <= 8192 ? 16u : ncells <= 65536 ? 4u : 1u; }
// bytes of the vertex copies (first part of the scratch)
inline size_t coarse_scratch_bytes(const cn_grid& g) {
  if (g.num_levels < 1) return 0;
  const unsigned n1 = coarse_n1(g);
  return n1 > COARSE_MAX_N1 ? 0 : (size_t)COARSE_COPIES * n1 * n1 * n1 * 2 * sizeof(float);
}
// the consecutive coarse levels that may be kept cell-major, and the bytes they need (second part of the scratch)
inline size_t cell_scratch_layout(const cn_grid& g, CellScatter* out) {
  CellScatter c{};
  unsigned long long floats = 0;
  for (int l = 0; l < g.num_levels && l < CN_CELL_LEVELS; ++l) {
    const unsigned n = cell_n(g, l);
    const unsigned long long cells = (unsigned long long)n * n * n;
    if (cells > CELL_MAX_CELLS) break;
    c.n[l] = n;
    c.copies[l] = cell_copies(cells);
    c.offset[l] = floats;
    floats += c.copies[l] * cells * 16ull;
    c.num_levels = l + 1;
  }
  if (out) *out = c;
  return (size_t)floats * sizeof(float);
}
// levels 0 .. k-1 with at most max_cells cells each (the launch passes ratio x samples).  Of a small level's copies only as
// many are used as the batch needs to keep the requests per record in the low hundreds: ~ samples / (48 cells), rounded up
// to a power of two.
inline CellScatter make_cell_scatter(const cn_grid& grads_grid, unsigned long long max_cells, unsigned long long samples) {
  CellScatter c{};
  if (!grads_grid.scatter_scratch) return c;
  const size_t head = coarse_scratch_bytes(grads_grid);
  const size_t need = cell_scratch_layout(grads_grid, &c);
  if (need == 0 || grads_grid.scatter_scratch_bytes < head) {
    c = CellScatter{};
    return c;
  }
  // the scratch may hold a PREFIX of the levels (cn_grid_scatter_scratch_bytes_for: sized for a maximum batch): use the
  // levels whose records fit it
  int fit = 0;
  while (fit < c.num_levels) {
    const unsigned long long cells = (unsigned long long)c.n[fit] * c.n[fit] * c.n[fit];
    const unsigned long long end = (c.offset[fit] + c.copies[fit] * cells * 16ull) * sizeof(float);
    if (head + end > grads_grid.scatter_scratch_bytes) break;
    ++fit;
  }
  c.num_levels = fit;
  int k = 0;
  while (k < c.num_levels && (unsigned long long)c.n[k] * c.n[k] * c.n[k] <= max_cells) ++k;
  c.num_levels = k;
  for (int l = 0; l < k; ++l) {
    const unsigned long long cells = (unsigned long long)c.n[l] * c.n[l] * c.n[l];
    unsigned want = 1;
    while (want < c.copies[l] && (unsigned long long)want * cells * 48ull < samples) want <<= 1;
    c.copies[l] = want;
  }
  c.base = k > 0 ? reinterpret_cast<float*>(static_cast<char*>(grads_grid.scatter_scratch) + head) : nullptr;
  return c;
}

__device__ __forceinline__ unsigned cell_n_of(const CellScatter& c, int l) {
  unsigned v = c.n[0];
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) v = l == k ? c.n[k] : v;
  return v;
}
__device__ __forceinline__ unsigned cell_copies_of(const CellScatter& c, int l) {
  unsigned v = c.copies[0];
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) v = l == k ? c.copies[k] : v;
  return v;
}
__device__ __forceinline__ unsigned long long cell_offset_of(const CellScatter& c, int l) {
  unsigned long long v = c.offset[0];
#pragma unroll
  for (int k = 1; k < CN_CELL_LEVELS; ++k) v = l == k ? c.offset[k] : v;
  return v;
}

// per-lane level record by static selects (a per-lane index into the kernarg arrays would go to scratch)
__device__ __forceinline__ Lvl lane_level(const GridDev& g, int l) {
  Lvl v = g.level(0);
#pragma unroll
  for (int k = 1; k < CN_MAX_LEVELS; ++k) {
    const Lvl w = g.level(k);
    v.off = l == k ? w.off : v.off;
    v.mask = l == k ? w.mask : v.mask;
    v.m1 = l == k ? w.m1 : v.m1;
    v.m2 = l == k ? w.m2 : v.m2;
    v.scale = l == k ? w.scale : v.scale;
  }
  return v;
}

// level records staged in LDS by a kernel's prologue: rec[0..16) scales, then [16][4] unsigned {off, mask, m1, m2}
constexpr int LVL_REC_FLOATS = 16 + 64;
__device__ __forceinline__ void lds_level_fill(float* rec, const GridDev& g, int tid) {
  if (tid < CN_MAX_LEVELS) {
    const Lvl v = lane_level(g, tid);
    rec[tid] = v.scale;
    unsigned* u = reinterpret_cast<unsigned*>(rec + 16 + 4 * tid);
    u[0] = v.off;
    u[1] = v.mask;
    u[2] = v.m1;
    u[3] = v.m2;
  }
}
__device__ __forceinline__ Lvl lds_level_rec(const float* rec, int l) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 r = *reinterpret_cast<const u32x4*>(rec + 16 + 4 * l);
  Lvl lv;
  lv.off = r.x;
  lv.mask = r.y;
  lv.m1 = r.z;
  lv.m2 = r.w;
  lv.scale = rec[l];
  return lv;
}

// grid dispatching on the table type (wave-uniform branch per level; the hot kernels are templated on it instead)
__device__ __forceinline__ float2 hash_level_any(const GridDev& g, int l, float px, float py, float pz) {
  const Lvl lv = g.level(l);
  return g.half ? hash_level<true>(g.table, lv, g.pos_offset, px, py, pz)
                : hash_level<false>(g.table, lv, g.pos_offset, px, py, pz);
}

// The same with the blend written per component.  hipcc then SLP-packs across corners (~18 extra v_mov per level and
// sample) but also interleaves each level's 8 gathers with the previous level's blend, waiting for them a few at a
// time -- in the render kernels' gather waves that schedule measures FASTER than the leaner code above (4.9 vs 4.4
// Gsamples/s at C2; hand-pipelining 16-32 gathers in flight per wave is slower still: 4.2; issuing a unit's 8 gathers
// together and blending them with the packed form, i.e. the same pacing with fewer instructions: 4.57 vs 4.84), so they
// keep this form.
template <bool HALF = false, bool OFFSET = true>
__device__ __forceinline__ float2 hash_level_sc(const void* __restrict__ table, const Lvl& lv, float pos_offset, float px,
                                                float py, float pz) {
  const Cell k = hash_cell<OFFSET>(lv, pos_offset, px, py, pz);
  const float ox = k.ox, oy = k.oy, oz = k.oz;
  float2 ccc = hash_gather<HALF>(table, ((k.hx1 ^ k.hy1 ^ k.hz1) & lv.mask) + lv.off);  // f_0
  float2 cfc = hash_gather<HALF>(table, ((k.hx1 ^ k.hy0 ^ k.hz1) & lv.mask) + lv.off);  // f_1
  float2 ffc = hash_gather<HALF>(table, ((k.hx0 ^ k.hy0 ^ k.hz1) & lv.mask) + lv.off);  // f_2
  float2 fcc = hash_gather<HALF>(table, ((k.hx0 ^ k.hy1 ^ k.hz1) & lv.mask) + lv.off);  // f_3
  float2 ccf = hash_gather<HALF>(table, ((k.hx1 ^ k.hy1 ^ k.hz0) & lv.mask) + lv.off);  // f_4
  float2 cff = hash_gather<HALF>(table, ((k.hx1 ^ k.hy0 ^ k.hz0) & lv.mask) + lv.off);  // f_5
  float2 fff = hash_gather<HALF>(table, ((k.hx0 ^ k.hy0 ^ k.hz0) & lv.mask) + lv.off);  // f_6
  float2 fcf = hash_gather<HALF>(table, ((k.hx0 ^ k.hy1 ^ k.hz0) & lv.mask) + lv.off);  // f_7
  float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
  float2 r;
  {
    float f03 = ccc.x * ox + fcc.x * mx, f12 = cfc.x * ox + ffc.x * mx;
    float f56 = cff.x * ox + fff.x * mx, f47 = ccf.x * ox + fcf.x * mx;
    float a = f03 * oy + f12 * my, b = f47 * oy + f56 * my;
    r.x = a * oz + b * mz;
  }
  {
    float f03 = ccc.y * ox + fcc.y * mx, f12 = cfc.y * ox + ffc.y * mx;
    float f56 = cff.y * ox + fff.y * mx, f47 = ccf.y * ox + fcf.y * mx;
    float a = f03 * oy + f12 * my, b = f47 * oy + f56 * my;
    r.y = a * oz + b * mz;
  }
  return r;
}

// Half table, fp16 matrix mode (CN_MATRIX_F16): the two features of an entry stay packed in one register and the trilinear
// blend runs on v_pk_add_f16 / v_pk_fma_f16 (14 packed instructions for both features instead of 28 fp32 ones plus 16
// conversions), interpolation weights rounded to fp16 -- the precision class of tcnn's kernel_grid, which accumulates the
// weighted corners in the parameter type (fp16).  Returns the packed (feature 0, feature 1) pair: one dword of the fp16 MFMA
// B operand, no conversion anywhere.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// The gathers are what this mode costs: a CU's L1 looks up ONE cache line per clock for per-lane-addressed loads whose
// lanes do not share lines (tools/gather_rate_microbench.hip: 64 cycles per wave-level load with 64 lines, whatever the
// width), and with fp16 MFMAs the render kernel runs at that rate (1.17e9 line lookups per C2 launch = 4.6e6 cycles per CU).
// So the two x-corners of a (y, z) row are fetched with ONE 8-byte load where they are neighbours: entries e and e ^ 1 (the
// aligned pair) whenever the cell's x index is even -- true for hashed levels (index = x ^ y*P1 ^ z*P2) and for the dense levels
// of the tcnn layout (x sits in the low bits) alike.  Lanes with an odd x index fetch their upper corner with a second,
// 4-byte load under the execution mask: 6 line lookups per level and sample instead of 8.
// Issued and blended in two steps so that a kernel can keep the next unit's loads in flight under this unit's blend (the
// conditional load is control flow: hipcc no longer schedules across it, and sinks every blend to the end of the gather
// phase -- 96 live registers and 370 bytes of scratch -- unless the order is written out, see pk_pin).
typedef unsigned u32x2p __attribute__((ext_vector_type(2)));
struct PkLoads {
  u32x2p pr[4];    // the aligned pair holding the lower-x corner of rows (y, z) = ff, cf, fc, cc
#ifdef CN_PK_UP_WIDE
  u32x2p up[4];    // odd x index: the aligned pair that holds the upper-x corner (its first entry: x + 1 is even)
#else
  unsigned up[4];  // the upper-x corner of each row, loaded separately when the x index is odd
#endif
  unsigned bits;   // bit r: the lower-x corner is the pair's second entry; bit 4: odd x index
  unsigned wxy, wzz;  // interpolation weights as packed fp16 pairs (x, y) and (z, z)
};
template <bool OFFSET = true>
__device__ __forceinline__ PkLoads hash_level_pk_issue(const void* __restrict__ table, const Lvl& lv, float pos_offset, float px,
                                                       float py, float pz) {
  const Cell k = hash_cell<OFFSET>(lv, pos_offset, px, py, pz);
  const char* base = reinterpret_cast<const char*>(table);
  const unsigned pair_mask = lv.mask & ~1u;
  const unsigned hyz[4] = {k.hy0 ^ k.hz0, k.hy1 ^ k.hz0, k.hy0 ^ k.hz1, k.hy1 ^ k.hz1};
  PkLoads L;
  {
    const _Float16 hx = (_Float16)k.ox, hy = (_Float16)k.oy, hz = (_Float16)k.oz;
    const f16x2 xy = {hx, hy}, zz = {hz, hz};
    L.wxy = __builtin_bit_cast(unsigned, xy);
    L.wzz = __builtin_bit_cast(unsigned, zz);
  }
  const bool odd = (k.hx0 & 1u) != 0u;
  L.bits = odd ? 16u : 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // level offsets are even, so the aligned pair of entry e is e & ~1
    const unsigned x = k.hx0 ^ hyz[r];
    L.bits |= (x & 1u) << r;
    L.pr[r] = *reinterpret_cast<const u32x2p*>(base + (size_t)(((x & pair_mask) + lv.off) << 2));
#ifdef CN_PK_UP_WIDE
    L.up[r] = u32x2p{0u, 0u};
#else
    L.up[r] = 0u;
#endif
  }
#if !defined(CN_ABLATE_X1)  // timing-only build: no second load
  if (odd) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#ifdef CN_PK_UP_WIDE  // 8-byte loads coalesce lane PAIRS (4-byte loads only whole quads): worth it when neighbouring lanes are
                      // neighbouring pixels
      L.up[r] = *reinterpret_cast<const u32x2p*>(base + (size_t)((((k.hx1 ^ hyz[r]) & pair_mask) + lv.off) << 2));
#else
      L.up[r] = *reinterpret_cast<const unsigned*>(base + (size_t)((((k.hx1 ^ hyz[r]) & lv.mask) + lv.off) << 2));
#endif
    }
  }
#endif
  return L;
}
__device__ __forceinline__ unsigned hash_level_pk_blend(const PkLoads& L) {
  const bool odd = (L.bits & 16u) != 0u;
  f16x2 lo[4], hi[4];  // lower-x / upper-x corner of each row
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool second = (L.bits & (1u << r)) != 0u;
    const unsigned a = L.pr[r].x, b = L.pr[r].y;
    lo[r] = __builtin_bit_cast(f16x2, second ? b : a);
#ifdef CN_PK_UP_WIDE
    // x + 1 is even: the upper corner's entry index has the parity of (x ^ y-term ^ z-term) ^ 1, i.e. it is the pair's
    // second entry exactly when the lower corner (odd x) was its pair's FIRST
    const unsigned upv = second ? L.up[r].x : L.up[r].y;
#else
    const unsigned upv = L.up[r];
#endif
    hi[r] = __builtin_bit_cast(f16x2, odd ? upv : (second ? a : b));
  }
  const f16x2 wxy = __builtin_bit_cast(f16x2, L.wxy), wz = __builtin_bit_cast(f16x2, L.wzz);
  const f16x2 wx = {wxy.x, wxy.x}, wy = {wxy.y, wxy.y};
  // lerp(lo, hi, w) = lo + (hi - lo) w: x toward the upper corner, then y, then z (the order of hash_level_sc)
  const f16x2 xff = lo[0] + (hi[0] - lo[0]) * wx, xcf = lo[1] + (hi[1] - lo[1]) * wx;
  const f16x2 xfc = lo[2] + (hi[2] - lo[2]) * wx, xcc = lo[3] + (hi[3] - lo[3]) * wx;
  const f16x2 yf = xff + (xcf - xff) * wy, yc = xfc + (xcc - xfc) * wy;
  const f16x2 r = yf + (yc - yf) * wz;
  return __builtin_bit_cast(unsigned, r);
}
// (The same blend on eight plain 4-byte gathers -- no pairs, no selects, no branch -- was tried for the proposal sampler's fp16 mode,
//  which is bound by VALU issue rather than by lookups: 0.646 vs 0.587 ms per launch with the pair form, so it is not kept.)
// keeps a blended value where the program computed it (an empty volatile asm is ordered against the others)
__device__ __forceinline__ unsigned pk_pin(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <bool OFFSET = true>
__device__ __forceinline__ unsigned hash_level_pk(const void* __restrict__ table, const Lvl& lv, float pos_offset, float px,
                                                  float py, float pz) {
  return hash_level_pk_blend(hash_level_pk_issue<OFFSET>(table, lv, pos_offset, px, py, pz));
}

// hipcc pitfall (ROCm 7.2), found while building a 16-byte "x-pair" gather variant (measured slower and removed:
// 3.69 vs 3.86 Gsamples/s): __builtin_bit_cast(float, v.y) applied directly to an ext_vector element reads element 0 --
// copy the element into a scalar first.

// torch.linspace(0,1,steps)[i] (CPU kernel: symmetric around the midpoint)
__device__ __forceinline__ float linspace01(int i, int steps) {
  float step = 1.f / (float)(steps - 1);
  int half = steps / 2;
  return i < half ? step * (float)i : 1.f - step * (float)(steps - i - 1);
}

__device__ __forceinline__ float spacing_fn(int mode, float x) {
  if (mode == CN_SPACING_PIECEWISE) return x < 1.f ? x / 2.f : 1.f - 1.f / (2.f * x);
  return x;
}
__device__ __forceinline__ float spacing_fn_inv(int mode, float x) {
  if (mode == CN_SPACING_PIECEWISE) return x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x);
  return x;
}
// spacing_to_euclidean_fn(x) = s^-1(x*s(far) + (1-x)*s(near))
__device__ __forceinline__ float spacing_to_euclid(int mode, float x, float s_near, float s_far) {
  return spacing_fn_inv(mode, x * s_far + (1.f - x) * s_near);
}

// nerfstudio components_from_spherical_harmonics(levels=4)
__device__ __forceinline__ void sh_deg4(float x, float y, float z, float* c) {
  float xx = x * x, yy = y * y, zz = z * z;
  c[0] = 0.28209479177387814f;
  c[1] = 0.4886025119029199f * y;
  c[2] = 0.4886025119029199f * z;
  c[3] = 0.4886025119029199f * x;
  c[4] = 1.0925484305920792f * x * y;
  c[5] = 1.0925484305920792f * y * z;
  c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  c[7] = 1.0925484305920792f * x * z;
  c[8] = 0.5462742152960396f * (xx - yy);
  c[9] = 0.5900435899266435f * y * (3.f * xx - yy);
  c[10] = 2.890611442640554f * x * y * z;
  c[11] = 0.4570457994644658f * y * (5.f * zz - 1.f);
  c[12] = 0.3731763325901154f * z * (5.f * zz - 3.f);
  c[13] = 0.4570457994644658f * x * (5.f * zz - 1.f);
  c[14] = 1.445305721320277f * z * (xx - yy);
  c[15] = 0.5900435899266435f * x * (xx - 3.f * yy);
}

__device__ __forceinline__ float nan_to_num(float v) {
  if (v != v) return 0.f;
  return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

}  // namespace cn

// Shared host/device helpers for libcropnerf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

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

inline unsigned grid_for(long long n, int block, long long cap = 1 << 20) {
  long long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------

struct GridDev {  // by-value kernel argument
  const float* table;
  int num_levels;
  unsigned mask;      // T-1
  unsigned level_stride;  // T
  float scale[CN_MAX_LEVELS];
};

inline GridDev make_grid_dev(const cn_grid& g) {
  GridDev d;
  d.table = g.table;
  d.num_levels = g.num_levels;
  d.level_stride = 1u << g.log2_table_size;
  d.mask = d.level_stride - 1u;
  for (int i = 0; i < CN_MAX_LEVELS; ++i) d.scale[i] = i < g.num_levels ? g.scalings[i] : 0.f;
  return d;
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

#define CN_P1 2654435761u
#define CN_P2 805459861u

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

// One level of the hash grid (HashEncoding.pytorch_fwd): 8 corner gathers of float2 + trilinear blend in the
// reference's order (x toward the ceil corner, then y, then z).
// `table` is the wave-uniform base of the whole [L*T,2] table and `level_off` = l*T the (possibly per-lane) level
// offset in entries: every gather address is base + 32-bit byte offset (global_load saddr+voffset form), so a gather
// costs one address VGPR and no 64-bit adds.  L*T*8 B <= 2^31 is checked on the host.
__device__ __forceinline__ float2 hash_gather(const float* __restrict__ table, unsigned entry) {
  return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(table) + (size_t)(entry << 3));
}

__device__ __forceinline__ float2 hash_level(const float* __restrict__ table, unsigned level_off, unsigned mask,
                                             float scale, float px, float py, float pz) {
  float sx = px * scale, sy = py * scale, sz = pz * scale;
  float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  float ox = sx - fx, oy = sy - fy, oz = sz - fz;
  unsigned ix = (unsigned)(int)fx, iy = (unsigned)(int)fy, iz = (unsigned)(int)fz;
  // ceil corner: ceil(s) == floor(s)+1 unless s is integral, where its weight (offset) is 0.
  unsigned hx0 = ix, hx1 = ix + 1u;
  unsigned hy0 = iy * CN_P1, hy1 = hy0 + CN_P1;
  unsigned hz0 = iz * CN_P2, hz1 = hz0 + CN_P2;
  float2 ccc = hash_gather(table, ((hx1 ^ hy1 ^ hz1) & mask) + level_off);  // f_0
  float2 cfc = hash_gather(table, ((hx1 ^ hy0 ^ hz1) & mask) + level_off);  // f_1
  float2 ffc = hash_gather(table, ((hx0 ^ hy0 ^ hz1) & mask) + level_off);  // f_2
  float2 fcc = hash_gather(table, ((hx0 ^ hy1 ^ hz1) & mask) + level_off);  // f_3
  float2 ccf = hash_gather(table, ((hx1 ^ hy1 ^ hz0) & mask) + level_off);  // f_4
  float2 cff = hash_gather(table, ((hx1 ^ hy0 ^ hz0) & mask) + level_off);  // f_5
  float2 fff = hash_gather(table, ((hx0 ^ hy0 ^ hz0) & mask) + level_off);  // f_6
  float2 fcf = hash_gather(table, ((hx0 ^ hy1 ^ hz0) & mask) + level_off);  // f_7
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

// The same with the blend written per component.  hipcc then SLP-packs across corners (~18 extra v_mov per level and
// sample) but also interleaves each level's 8 gathers with the previous level's blend, waiting for them a few at a
// time -- in the render kernels' gather waves that schedule measures FASTER than the leaner code above (4.9 vs 4.4
// Gsamples/s at C2; hand-pipelining 16-32 gathers in flight per wave is slower still: 4.2; issuing a unit's 8 gathers
// together and blending them with the packed form, i.e. the same pacing with fewer instructions: 4.57 vs 4.84), so they
// keep this form.
__device__ __forceinline__ float2 hash_level_sc(const float* __restrict__ table, unsigned level_off, unsigned mask,
                                             float scale, float px, float py, float pz) {
  float sx = px * scale, sy = py * scale, sz = pz * scale;
  float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  float ox = sx - fx, oy = sy - fy, oz = sz - fz;
  unsigned ix = (unsigned)(int)fx, iy = (unsigned)(int)fy, iz = (unsigned)(int)fz;
  // ceil corner: ceil(s) == floor(s)+1 unless s is integral, where its weight (offset) is 0.
  unsigned hx0 = ix, hx1 = ix + 1u;
  unsigned hy0 = iy * CN_P1, hy1 = hy0 + CN_P1;
  unsigned hz0 = iz * CN_P2, hz1 = hz0 + CN_P2;
  float2 ccc = hash_gather(table, ((hx1 ^ hy1 ^ hz1) & mask) + level_off);  // f_0
  float2 cfc = hash_gather(table, ((hx1 ^ hy0 ^ hz1) & mask) + level_off);  // f_1
  float2 ffc = hash_gather(table, ((hx0 ^ hy0 ^ hz1) & mask) + level_off);  // f_2
  float2 fcc = hash_gather(table, ((hx0 ^ hy1 ^ hz1) & mask) + level_off);  // f_3
  float2 ccf = hash_gather(table, ((hx1 ^ hy1 ^ hz0) & mask) + level_off);  // f_4
  float2 cff = hash_gather(table, ((hx1 ^ hy0 ^ hz0) & mask) + level_off);  // f_5
  float2 fff = hash_gather(table, ((hx0 ^ hy0 ^ hz0) & mask) + level_off);  // f_6
  float2 fcf = hash_gather(table, ((hx0 ^ hy1 ^ hz0) & mask) + level_off);  // f_7
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

// hash_level with the x-neighbour pairs fetched together where the table layout allows it (device-only, gfx950).
// The two corners that differ only in x hash to e and e' = e ^ (ix ^ (ix+1)); for even ix that is e ^ 1, i.e. the
// two entries share one aligned 16-byte slot, so ONE 16-byte gather returns both (4 instead of 8 lane-requests per
// sample and level); for odd ix the ceil-x corner needs its own 8-byte gather.  On average 6 instead of 8
// lane-requests -- the fused kernel is bound by the rate at which the texture-address unit takes divergent lanes.
//
// The gathers are buffer loads through a resource (V#) over the whole table: lanes that do not need the second
// gather present an out-of-range offset, which the range check drops before the cache -- predication without a
// divergent branch (an `if (odd)` around the loads makes hipcc spill ~700 B per lane in the fused kernel).
//
// hipcc pitfall (ROCm 7.2): __builtin_bit_cast(float, v.y) on an ext_vector element reads element 0 -- always copy
// the element into a scalar first (as_f32 below takes it by value).
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float as_f32(unsigned v) { return __builtin_bit_cast(float, v); }

// The loads are issued from inline asm so that exactly 4 x 16-byte + 4 x 8-byte gathers per sample and level reach
// the memory pipe, in that order, with the registers we choose (through the raw_buffer_load builtins hipcc hoists all
// eight levels' loads and spills ~450 B per lane in the fused kernel).  Inline-asm loads are invisible to the compiler's
// vmcnt accounting, so the results pass through xpair_wait() (an `s_waitcnt vmcnt(0)` that takes the destination
// registers as in/out operands) before anything reads them.
__device__ __forceinline__ u32x4_t table_rsrc(const float* table, unsigned bytes) {
  const unsigned long long a = (unsigned long long)table;
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);  // the descriptor must sit in SGPRs
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);  // base[47:32], stride 0
  r[2] = __builtin_amdgcn_readfirstlane(bytes);                          // num_records (bytes for a raw buffer)
  r[3] = 0x00020000u;                                                    // DATA_FORMAT = 32
  return r;
}

struct XPairLoads {
  u32x4_t pair[4];   // aligned slot holding the floor-x corner (and, for even ix, the ceil-x corner)
  u32x2_t extra[4];  // ceil-x corner when ix is odd (zeros otherwise)
};

__device__ __forceinline__ void xpair_issue(XPairLoads& L, u32x4_t rsrc, unsigned level_off, unsigned mask,
                                            float scale, float px, float py, float pz) {
  const unsigned ix = (unsigned)(int)floorf(px * scale), iy = (unsigned)(int)floorf(py * scale),
                 iz = (unsigned)(int)floorf(pz * scale);
  const unsigned hy0 = iy * CN_P1, hy1 = hy0 + CN_P1;
  const unsigned hz0 = iz * CN_P2, hz1 = hz0 + CN_P2;
  const unsigned yz[4] = {hy1 ^ hz1, hy0 ^ hz1, hy1 ^ hz0, hy0 ^ hz0};  // (c,c) (f,c) (c,f) (f,f) in (y,z)
  const bool odd = (ix & 1u) != 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned off = ((((ix ^ yz[k]) & mask) & ~1u) + level_off) << 3;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(L.pair[k]) : "v"(off), "s"(rsrc) : "memory");
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned off = odd ? ((((ix + 1u) ^ yz[k]) & mask) + level_off) << 3 : 0xfffffff0u;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(L.extra[k]) : "v"(off), "s"(rsrc) : "memory");
  }
}

// every read of A / B must come after this
__device__ __forceinline__ void xpair_wait(XPairLoads& A, XPairLoads& B) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(A.pair[0]), "+v"(A.pair[1]), "+v"(A.pair[2]), "+v"(A.pair[3]), "+v"(A.extra[0]), "+v"(A.extra[1]),
                 "+v"(A.extra[2]), "+v"(A.extra[3]), "+v"(B.pair[0]), "+v"(B.pair[1]), "+v"(B.pair[2]), "+v"(B.pair[3]),
                 "+v"(B.extra[0]), "+v"(B.extra[1]), "+v"(B.extra[2]), "+v"(B.extra[3])
               :
               : "memory");
}

__device__ __forceinline__ float2 xpair_blend(const XPairLoads& L, float scale, float px, float py, float pz) {
  float sx = px * scale, sy = py * scale, sz = pz * scale;
  float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  float ox = sx - fx, oy = sy - fy, oz = sz - fz;
  unsigned ix = (unsigned)(int)fx, iy = (unsigned)(int)fy, iz = (unsigned)(int)fz;
  const unsigned hy0 = iy * CN_P1, hy1 = hy0 + CN_P1;
  const unsigned hz0 = iz * CN_P2, hz1 = hz0 + CN_P2;
  const unsigned yz[4] = {hy1 ^ hz1, hy0 ^ hz1, hy1 ^ hz0, hy0 ^ hz0};
  const bool odd = (ix & 1u) != 0u;
  float2 lo[4], hi[4];  // floor-x / ceil-x corner of each (y,z) combination
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool upper = ((ix ^ yz[k]) & 1u) != 0u;  // which half of the slot is the floor-x entry (mask keeps bit 0)
    const float ax = as_f32(L.pair[k][0]), ay = as_f32(L.pair[k][1]);
    const float bx = as_f32(L.pair[k][2]), by = as_f32(L.pair[k][3]);
    lo[k].x = upper ? bx : ax;
    lo[k].y = upper ? by : ay;
    hi[k].x = odd ? as_f32(L.extra[k][0]) : (upper ? ax : bx);
    hi[k].y = odd ? as_f32(L.extra[k][1]) : (upper ? ay : by);
  }
  // reference blend order: x toward the ceil corner, then y, then z  (f_0..f_7 = ccc cfc ffc fcc ccf cff fff fcf)
  const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
  float2 r;
  {
    float f03 = hi[0].x * ox + lo[0].x * mx, f12 = hi[1].x * ox + lo[1].x * mx;
    float f47 = hi[2].x * ox + lo[2].x * mx, f56 = hi[3].x * ox + lo[3].x * mx;
    float a = f03 * oy + f12 * my, b = f47 * oy + f56 * my;
    r.x = a * oz + b * mz;
  }
  {
    float f03 = hi[0].y * ox + lo[0].y * mx, f12 = hi[1].y * ox + lo[1].y * mx;
    float f47 = hi[2].y * ox + lo[2].y * mx, f56 = hi[3].y * ox + lo[3].y * mx;
    float a = f03 * oy + f12 * my, b = f47 * oy + f56 * my;
    r.y = a * oz + b * mz;
  }
  return r;
}

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

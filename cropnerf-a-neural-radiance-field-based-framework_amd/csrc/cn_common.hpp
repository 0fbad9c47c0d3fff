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

// Wave64 cross-lane primitives used by the compositor and the samplers (gfx950: 64 lanes, always).
// DPP / permlane-swap / readlane forms only: none of them needs a per-lane address register (ds_bpermute does, and
// the compiler hoists those addresses out of the ray loop where they pin VGPRs for the whole kernel).
#pragma once

#include <hip/hip_runtime.h>

namespace cn {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// DPP controls (ISA: DPP_CTRL)
#define CN_DPP_ROW_SHR(n) (0x110 + (n))
#define CN_DPP_ROW_BCAST15 0x142
#define CN_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK, bool BOUND_CTRL>
__device__ __forceinline__ float dpp_zero(float v) {  // permuted v; lanes without a source read 0
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND_CTRL));
}

// inclusive prefix sum over the 64 lanes (lane order): Hillis-Steele inside each row of 16, then the two
// cross-row carries (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3).
__device__ __forceinline__ float wave_inclusive_scan(float v) {
  v += dpp_zero<CN_DPP_ROW_SHR(1), 0xf, true>(v);
  v += dpp_zero<CN_DPP_ROW_SHR(2), 0xf, true>(v);
  v += dpp_zero<CN_DPP_ROW_SHR(4), 0xf, true>(v);
  v += dpp_zero<CN_DPP_ROW_SHR(8), 0xf, true>(v);
  v += dpp_zero<CN_DPP_ROW_BCAST15, 0xa, false>(v);
  v += dpp_zero<CN_DPP_ROW_BCAST31, 0xc, false>(v);
  return v;
}

// value of `v` in lane `src` (src must be wave-uniform) -> every lane
__device__ __forceinline__ float wave_read(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__device__ __forceinline__ float wave_sum(float v) { return wave_read(wave_inclusive_scan(v), 63); }

// v_permlane16_swap / v_permlane32_swap (gfx950).  hipcc pitfall (ROCm 7.2): __builtin_bit_cast(float, r[1]) applied
// directly to an ext_vector element reads element 0 (a lo+hi sum silently becomes 2*lo) -- the elements are copied
// into scalars first.
//   swap16: rows 1,3 of a <-> rows 0,2 of b      swap32: lanes 32..63 of a <-> lanes 0..31 of b
__device__ __forceinline__ void permlane16_swap(unsigned& a, unsigned& b) {
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = r0;
  b = r1;
}
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = r0;
  b = r1;
}

// sum over lanes {l, l^16, l^32, l^48}
__device__ __forceinline__ float rows_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);  // a = {r0,r0,r2,r2}, b = {r1,r1,r3,r3}
  float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  a = __builtin_bit_cast(unsigned, s);
  b = a;
  permlane32_swap(a, b);  // a = {lo,lo}, b = {hi,hi}
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// copy row 0 (lanes 0..15) of v into every row: lane (g, j) <- lane j
__device__ __forceinline__ float row0_broadcast(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);  // a = {r0, r0, r2, r2}
  b = a;
  permlane32_swap(a, b);  // a = {r0, r0, r0, r0}
  return __builtin_bit_cast(float, a);
}

}  // namespace cn

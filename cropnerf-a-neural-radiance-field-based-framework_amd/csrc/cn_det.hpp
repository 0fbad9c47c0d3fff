// Deterministic gradient accumulation (a TEST build: libcropnerf_hip_det.so, -DCN_DETERMINISTIC_SCATTER=1).
//
// The training kernels sum gradients, loss terms and per-camera pose gradients with float atomics; the order in which the
// memory side serves them changes from run to run, and with it the rounding of every sum.  Adam's normalisation amplifies
// that (a rarely-hit table entry moves by ~lr whatever the size of its gradient), so two runs of the same iteration differ
// by per cents after a few steps and a test can only bound them loosely.  In this build every global float atomic of the
// training kernels -- cn_atomic_add below -- adds round(v * 2^44) to a 64-bit integer SHADOW of its destination instead
// (integer addition is associative: any order gives the same sum), and cn::det_flush, enqueued behind every kernel that
// may have accumulated, adds the shadows to the float destinations in one fixed-order pass and zeroes them.  Tile-to-
// workgroup assignment is static in every kernel and the in-register / DPP / MFMA partial sums are order-fixed, so two
// runs of the same launch sequence produce the same bits.  Resolution 2^-44 = 5.7e-14, range +-5.2e5 per entry: not the
// default build's arithmetic (a gradient below the resolution vanishes), which is why this is a separate library that only
// tests load (CN_DETERMINISTIC_SCATTER=1 selects it in cropnerf_amd/_lib.py).  Speed is not a goal: every destination is
// looked up in a small table of registered ranges (cn_deterministic_register), and a flush walks every registered range.
//
// A destination that is not inside a registered range falls back to the float atomic and counts a MISS
// (cn_deterministic_misses): the tests assert zero.  Atomics on LDS keep their float form; the three places where several
// waves add to one LDS word (the block fold, the per-camera pose sums) switch to their global-memory forms in this build.
#pragma once
#include <hip/hip_runtime.h>

#ifndef CN_DETERMINISTIC_SCATTER
#define CN_DETERMINISTIC_SCATTER 0
#endif

namespace cn {

#if CN_DETERMINISTIC_SCATTER

constexpr int DET_MAX_RANGES = 16;
constexpr double DET_SCALE = 17592186044416.0;  // 2^44
struct DetRange {
  float* base;
  unsigned long long count;  // floats
  long long* shadow;         // [count]
};
struct DetTable {
  int n;
  unsigned long long* misses;  // device counter (may be null)
  DetRange r[DET_MAX_RANGES];
};

// one copy per translation unit (no relocatable device code in this build): deterministic.hip keeps the master table on
// the host and calls every unit's uploader when it changes
static __device__ DetTable det_table;
void det_add_uploader(int (*fn)(const DetTable*));
int det_flush(hipStream_t stream);  // 0 or a cn_status
static int det_upload_this_unit(const DetTable* t) {
  return hipMemcpyToSymbol(HIP_SYMBOL(det_table), t, sizeof(DetTable), 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -3;
}
static const int det_uploader_registered = (det_add_uploader(det_upload_this_unit), 0);

__device__ __forceinline__ void cn_atomic_add(float* p, float v) {
  const int n = det_table.n;
  for (int i = 0; i < n; ++i) {
    float* base = det_table.r[i].base;
    const unsigned long long off = (unsigned long long)(p - base);
    if (p >= base && off < det_table.r[i].count) {
      const long long q = __double2ll_rn((double)v * DET_SCALE);
      atomicAdd(reinterpret_cast<unsigned long long*>(det_table.r[i].shadow + off), (unsigned long long)q);
      return;
    }
  }
  if (det_table.misses) atomicAdd(det_table.misses, 1ull);
  atomicAdd(p, v);
}
#define CN_DET_FLUSH(stream)                              \
  do {                                                    \
    if (int cn_det_rc_ = cn::det_flush(stream)) return cn_det_rc_; \
  } while (0)

#else

__device__ __forceinline__ void cn_atomic_add(float* p, float v) { atomicAdd(p, v); }
#define CN_DET_FLUSH(stream) ((void)0)

#endif

}  // namespace cn

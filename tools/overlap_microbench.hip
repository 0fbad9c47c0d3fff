// Do MFMA passes and VALU instructions of DIFFERENT waves overlap on one SIMD?  (gfx950)
// hipcc -O3 --offload-arch=gfx950 tools/overlap_microbench.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k(int mode, int iters, float* out) {
  // 512 threads = 8 waves = 2 per SIMD: waves 0-3 (one per SIMD) role A, waves 4-7 role B
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float v0 = threadIdx.x, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
  const bool do_mfma = (mode == 0) || (mode == 2 && roleA) || (mode == 3);
  const bool do_valu = (mode == 1) || (mode == 2 && !roleA) || (mode == 4);
  // mode 0: all waves MFMA; 1: all waves VALU; 2: A = MFMA, B = VALU; 3: only A waves MFMA (B idle); 4: only B waves VALU
  if ((mode == 3 && !roleA) || (mode == 4 && roleA)) return;
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, v2, acc, 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {  // 4 independent chains of v_fma_f32
        v0 = __builtin_fmaf(v0, v1, v2);
        v1 = __builtin_fmaf(v1, v2, v3);
        v2 = __builtin_fmaf(v2, v3, v0);
        v3 = __builtin_fmaf(v3, v0, v1);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + v0 + v1 + v2 + v3;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const char* names[] = {"all 8 waves MFMA (32 per iter)", "all 8 waves VALU (256 fma per iter)", "4 waves MFMA + 4 waves VALU",
                         "4 waves MFMA only", "4 waves VALU only"};
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<<<256, 512>>>(mode, 100, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<256, 512>>>(mode, 2000, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-40s %8.3f ms\n", names[mode], ms);
  }
  return 0;
}

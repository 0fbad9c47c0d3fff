// Does a tile loop that starts every tile with synchronous random gathers slow down under a concurrent atomics kernel,
// and does issuing the gathers one tile ahead fix it?  (one 512-thread, 159 KB-LDS workgroup per CU, as field_backward_mfma)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>  // 0: no gathers, 1: synchronous gathers at the tile start, 2: gathers issued one tile ahead
__global__ void __launch_bounds__(512) heavy(int iters, int mf, const float2* table, unsigned mask, float* out) {
  extern __shared__ float lds[];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 99u;
  float2 pre[8];
  if (MODE == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { h = mix(h + j); pre[j] = table[h & mask]; }
  }
  for (int it = 0; it < iters; ++it) {
    float g = 0.f;
    if (MODE == 1) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { h = mix(h + j); v[j] = table[h & mask]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) g += v[j].x + v[j].y;
    }
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g += pre[j].x + pre[j].y;   // waits for the loads issued a tile ago
#pragma unroll
      for (int j = 0; j < 8; ++j) { h = mix(h + j); pre[j] = table[h & mask]; }
    }
    lds[(threadIdx.x * 17 + it) & 8191] = a + g;
    __syncthreads();
    for (int j = 0; j < mf; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    a = lds[(threadIdx.x * 5 + it) & 8191] + acc[0] * 1e-30f;
    __syncthreads();
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + a;
}
__global__ void __launch_bounds__(256) scatter(float* table, unsigned slot_mask, int iters) {
  const unsigned quad = (blockIdx.x * 256u + threadIdx.x) >> 2, ql = threadIdx.x & 3;
  unsigned h = quad * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    atomicAdd(table + 4 * (size_t)(h & slot_mask) + ql, 1.0f);
  }
}
int main() {
  float *out, *table, *gtab;
  (void)hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
  (void)hipMalloc(&table, 64u << 20); (void)hipMemset(table, 0, 64u << 20);
  (void)hipMalloc(&gtab, 64u << 20); (void)hipMemset(gtab, 0, 64u << 20);
  (void)hipFuncSetAttribute((const void*)heavy<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  (void)hipFuncSetAttribute((const void*)heavy<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  (void)hipFuncSetAttribute((const void*)heavy<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  hipStream_t s1, s2;
  (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const unsigned amask = (64u << 20) / 16 - 1, gmask = (64u << 20) / 8 - 1;
  const int iters = 400, mf = 256;  // 256 MFMAs x 32 cycles x 2 waves per SIMD = 16 k cycles per tile
  auto run = [&](int mode, int scat) {
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    if (mode == 0) hipLaunchKernelGGL(heavy<0>, dim3(256), dim3(512), 159 * 1024, s1, iters, mf, (const float2*)gtab, gmask, out);
    if (mode == 1) hipLaunchKernelGGL(heavy<1>, dim3(256), dim3(512), 159 * 1024, s1, iters, mf, (const float2*)gtab, gmask, out);
    if (mode == 2) hipLaunchKernelGGL(heavy<2>, dim3(256), dim3(512), 159 * 1024, s1, iters, mf, (const float2*)gtab, gmask, out);
    if (scat) hipLaunchKernelGGL(scatter, dim3(256 * 8), dim3(256), 0, s2, table, amask, scat);
    (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
  };
  run(1, 64);
  const char* names[] = {"no gathers", "synchronous gathers", "gathers one tile ahead"};
  for (int rep = 0; rep < 2; ++rep) {
    const float sc = run(-1, 300);
    for (int mode = 0; mode < 3; ++mode) {
      const float alone = run(mode, 0), both = run(mode, 300);
      printf("%-24s alone %.3f ms   scatter alone %.3f ms   both %.3f ms   (sum %.3f)\n", names[mode], alone, sc, both, alone + sc);
    }
  }
  return 0;
}

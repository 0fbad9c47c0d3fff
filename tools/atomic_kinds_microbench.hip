// Rates of different atomic kinds / plain stores / occupancies with the quad pattern (random 16-byte slots), MI355X.
// hipcc -O3 --offload-arch=gfx950 tools/atomic_kinds_microbench.hip -o /tmp/atomic_kinds && /tmp/atomic_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int KIND>
__global__ void __launch_bounds__(256) k(void* table, unsigned slot_mask, int iters) {
  const unsigned quad = (blockIdx.x * 256u + threadIdx.x) >> 2, gl = threadIdx.x & 3u;
  unsigned h = quad * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    const size_t slot = h & slot_mask;
    if (KIND == 0) atomicAdd((float*)table + 4 * slot + gl, 1.0f);
    if (KIND == 1) atomicAdd((unsigned*)table + 4 * slot + gl, 1u);
    if (KIND == 2) atomicAdd((unsigned long long*)table + 4 * slot + gl, 1ull);   // 32-byte slot
    if (KIND == 3) atomicAdd((double*)table + 4 * slot + gl, 1.0);                 // 32-byte slot
    if (KIND == 4) ((float*)table)[4 * slot + gl] = 1.0f;                          // plain store
    if (KIND == 5) atomicMax((unsigned*)table + 4 * slot + gl, h);
    if (KIND == 6) { if (gl < 2) atomicAdd((unsigned long long*)table + 2 * slot + gl, 1ull); }  // 2 lanes x 8 B = the same 16 B
    if (KIND == 7) { if (gl == 0) { float4 v = {1, 1, 1, 1}; *(float4*)((float*)table + 4 * slot) = v; } }  // one 16-byte store per quad
  }
}
int main() {
  void* table; (void)hipMalloc(&table, 1024ull << 20); (void)hipMemset(table, 0, 1024ull << 20);
  const char* names[] = {"f32 add", "u32 add", "u64 add (32 B slot)", "f64 add (32 B slot)", "plain 4-byte stores", "u32 max", "u64 add, 2 lanes = 16 B", "one 16-byte store per quad"};
  for (int blocks : {256 * 8, 256 * 2, 256}) {
    for (size_t mb : {4, 256}) {
      for (int kind = 0; kind < 8; ++kind) {
        const unsigned slot_mask = (unsigned)((mb << 20) / 32) - 1;
        const int iters = 256;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        auto launch = [&]() {
          switch (kind) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
            case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, iters); break;
          }
        };
        launch();
        (void)hipEventRecord(a, 0);
        for (int r = 0; r < 3; ++r) launch();
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("blocks %5d table %4zu MB  %-28s %8.3f ms  %7.2f G quads/s\n", blocks, mb, names[kind], ms, (double)blocks * 64 * iters / ms * 1e-6);
      }
    }
  }
  return 0;
}

// Do float atomics that stay inside one XCD's L2 run faster than device-scope ones on MI355X (gfx950)?
// hipcc -O3 --offload-arch=gfx950 tools/atomic_scope_microbench.hip -o /tmp/atomic_scope && /tmp/atomic_scope
// A device-scope atomic (sc1) is performed where all 8 L2s agree -- the memory side; a workgroup-scope one (no sc1) is
// performed by the L2 of the XCD the wave runs on.  The second kind is only CORRECT when no other XCD touches the line
// during the kernel, i.e. when every XCD accumulates into its own copy (picked by the hardware XCC_ID, not by blockIdx).
// Every lane adds 1.0f; the 4 lanes of a quad hit the 4 floats of one random 16-byte slot (hash_level_backward's shape).
// Reported: time, quads/s, and the sum over the table against the number of adds (lost updates show up as a deficit).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned xcc_id() {
  return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
}
template <int SCOPE>
__global__ void __launch_bounds__(256) k(float* table, unsigned slot_mask, int own, int iters, unsigned* xcc_hist) {
  const unsigned quad = (blockIdx.x * 256u + threadIdx.x) >> 2, gl = threadIdx.x & 3u;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(xcc_hist + 16 * (blockIdx.x & 7) + x, 1u);
  float* base = table + (own ? 4 * (size_t)x * (slot_mask + 1) : 0);
  unsigned h = quad * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    float* p = base + 4 * (size_t)(h & slot_mask) + gl;
    __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}
__global__ void sum_k(const float* t, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) s += t[i];
  atomicAdd(out, s);
}
int main() {
  const size_t max_bytes = 1024ull << 20;
  float* table; double* total; unsigned* hist;
  (void)hipMalloc(&table, max_bytes); (void)hipMalloc(&total, 8); (void)hipMalloc(&hist, 128 * 4);
  const int blocks = 256 * 8, iters = 256;
  struct Case { const char* name; size_t part_bytes; int own; int scope; };
  const Case cases[] = {
      {"agent scope, shared 4 MB", 4u << 20, 0, 0},       {"agent scope, shared 64 MB", 64u << 20, 0, 0},
      {"agent scope, per-XCD 8 x 4 MB", 4u << 20, 1, 0},  {"agent scope, per-XCD 8 x 64 MB", 64u << 20, 1, 0},
      {"workgroup scope, per-XCD 8 x 512 KB", 512u << 10, 1, 1}, {"workgroup scope, per-XCD 8 x 4 MB", 4u << 20, 1, 1},
      {"workgroup scope, per-XCD 8 x 16 MB", 16u << 20, 1, 1},   {"workgroup scope, per-XCD 8 x 64 MB", 64u << 20, 1, 1},
      {"wavefront scope, per-XCD 8 x 64 MB", 64u << 20, 1, 2},
      {"workgroup scope, SHARED 64 MB (wrong by design)", 64u << 20, 0, 1},
  };
  for (const Case& c : cases) {
    const unsigned slot_mask = (unsigned)(c.part_bytes / 16) - 1;
    const size_t bytes = c.part_bytes * (c.own ? 8 : 1);
    (void)hipMemset(table, 0, bytes); (void)hipMemset(total, 0, 8); (void)hipMemset(hist, 0, 512);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto launch = [&]() {
      if (c.scope == 0) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, c.own, iters, hist);
      else if (c.scope == 1) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, c.own, iters, hist);
      else hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WAVEFRONT>, dim3(blocks), dim3(256), 0, 0, table, slot_mask, c.own, iters, hist);
    };
    launch();
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 3;
    hipLaunchKernelGGL(sum_k, dim3(1024), dim3(256), 0, 0, table, bytes / 4, total);
    double got; (void)hipMemcpy(&got, total, 8, hipMemcpyDeviceToHost);
    const double adds = 4.0 * blocks * 256 * iters;
    const double quads = (double)blocks * 64 * iters;
    printf("%-50s %8.3f ms  %7.2f G quads/s   sum/adds = %.6f\n", c.name, ms, quads / ms * 1e-6, got / adds);
  }
  std::vector<unsigned> h(128);
  (void)hipMemcpy(h.data(), hist, 512, hipMemcpyDeviceToHost);
  printf("XCC_ID seen by workgroups with blockIdx %% 8 = r (last case):\n");
  for (int r = 0; r < 8; ++r) { printf("  r=%d:", r); for (int x = 0; x < 16; ++x) if (h[16 * r + x]) printf(" xcc%d x%u", x, h[16 * r + x]); printf("\n"); }
  return 0;
}

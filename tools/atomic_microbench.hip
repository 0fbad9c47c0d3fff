// What bounds float atomics into a hash-grid gradient table on MI355X?  (gfx950)
// hipcc -O3 --offload-arch=gfx950 tools/atomic_microbench.hip -o /tmp/atomic && /tmp/atomic
// Every lane adds one float; the G lanes of a group hit the G floats of one random, aligned G-float slot (G = 4 is the
// request shape of hash_level_backward).  Modes:
//   shared     every workgroup draws slots from the whole table (all 8 XCDs touch every line)
//   xcd-own    workgroup b draws from part (b % 8) of the table (round-robin dispatch: b % 8 = XCD), so a line is only
//              ever touched through ONE L2
//   lds        same draws, accumulated with LDS atomics into a 64 KB window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void __launch_bounds__(256) k(float* table, unsigned slot_mask, int parts, int iters, int mode, int log2g, int active) {
  __shared__ float win[16384];
  if (mode == 2) {
    for (int i = threadIdx.x; i < 16384; i += 256) win[i] = 0.f;
    __syncthreads();
  }
  const unsigned g = 1u << log2g;
  const unsigned group = (blockIdx.x * 256u + threadIdx.x) >> log2g, gl = threadIdx.x & (g - 1);
  const unsigned part = parts > 1 ? blockIdx.x % parts : 0;
  unsigned h = group * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    h = mix(h + it);
    const unsigned slot = h & slot_mask;
    if ((int)gl >= active) continue;
    if (mode == 2) atomicAdd(win + ((g * slot + gl) & 16383u), 1.0f);
    else if (mode == 3) {  // quad = two 8-byte pairs at random places of ONE 64-byte segment
      const unsigned seg = h & (slot_mask >> 2), p0 = (h >> 24) & 7u, p1 = (p0 + 1 + ((h >> 27) % 7u)) & 7u;
      atomicAdd(table + 16 * (size_t)seg + 2 * ((gl & 2) ? p1 : p0) + (gl & 1), 1.0f);
    } else if (mode == 4) {  // quad = two 8-byte pairs in two different segments
      const unsigned seg = (h ^ ((gl & 2) ? 0x5bd1e995u : 0u)) & (slot_mask >> 2), p0 = (h >> 24) & 7u;
      atomicAdd(table + 16 * (size_t)seg + 2 * p0 + (gl & 1), 1.0f);
    } else if (mode == 5) {  // the four floats of a slot come from lanes 16 apart
      const unsigned l = threadIdx.x & 63u, src = (blockIdx.x * 256u + (threadIdx.x & ~63u)) / 4 + (l & 15u);
      const unsigned hs = mix(src * 2654435761u + 12345u + it);
      atomicAdd(table + 4 * (size_t)(hs & slot_mask) + (l >> 4), 1.0f);
    } else if (mode == 6) {  // all four lanes of a quad hit the SAME float
      atomicAdd(table + 4 * (size_t)slot, 1.0f);
    } else atomicAdd(table + g * ((size_t)part * (slot_mask + 1) + slot) + gl, 1.0f);
  }
  if (mode == 2) {
    __syncthreads();
    for (int i = threadIdx.x; i < 16384; i += 256) table[i] += win[i];
  }
}
// random 8-byte gathers (float2, the hash grid's read) for comparison: one 64-byte segment request per lane
__global__ void __launch_bounds__(256) gather_k(const float2* table, unsigned entry_mask, int iters, float* out) {
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 777u;
  float acc = 0.f;
  for (int it = 0; it < iters; it += 4) {
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h = mix(h + it + j);
      v[j] = table[h & entry_mask];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y;
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const size_t max_bytes = 512u << 20;
  float* table;
  (void)hipMalloc(&table, max_bytes);
  (void)hipMemset(table, 0, max_bytes);
  const int blocks = 256 * 8, iters = 256;
  struct Case { const char* name; size_t part_bytes; int parts; int mode; int log2g; int active; };
  const Case cases[] = {
      {"shared 4 MB,  G=4", 4u << 20, 1, 0, 2, 4},   {"shared 64 MB, G=4", 64u << 20, 1, 0, 2, 4},
      {"xcd-own 8 x 512 KB, G=4", 512u << 10, 8, 1, 2, 4}, {"xcd-own 8 x 8 MB, G=4", 8u << 20, 8, 1, 2, 4},
      {"shared 64 MB, G=1", 64u << 20, 1, 0, 0, 1},  {"shared 64 MB, G=2", 64u << 20, 1, 0, 1, 2},
      {"shared 64 MB, G=4, 1 lane active", 64u << 20, 1, 0, 2, 1}, {"shared 64 MB, G=4, 2 lanes active", 64u << 20, 1, 0, 2, 2},
      {"shared 64 MB, G=8", 64u << 20, 1, 0, 3, 8},  {"shared 64 MB, G=16", 64u << 20, 1, 0, 4, 16},
      {"shared 64 MB, G=32", 64u << 20, 1, 0, 5, 32}, {"shared 64 MB, G=64", 64u << 20, 1, 0, 6, 64},
      {"two pairs, same 64 B segment", 64u << 20, 1, 3, 2, 4}, {"two pairs, different segments", 64u << 20, 1, 4, 2, 4},
      {"quad from lanes 16 apart", 64u << 20, 1, 5, 2, 4}, {"quad on one float (conflict)", 64u << 20, 1, 6, 2, 4},
      {"lds window, G=4", 4u << 20, 1, 2, 2, 4},     {"lds window, G=1", 4u << 20, 1, 2, 0, 1},
      {"lds window, G=64", 4u << 20, 1, 2, 6, 64},
  };
  for (const Case& c : cases) {
    const unsigned g = 1u << c.log2g;
    const unsigned slot_mask = (unsigned)(c.part_bytes / (4 * g)) - 1;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, slot_mask, c.parts, iters, c.mode, c.log2g, c.active);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, slot_mask, c.parts, iters, c.mode, c.log2g, c.active);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double groups = (double)blocks * 256 / g * iters;
    printf("%-36s %8.3f ms  %7.2f G groups/s  %7.1f G floats/s\n", c.name, ms, groups / ms * 1e-6, c.active * groups / ms * 1e-6);
  }
  for (size_t bytes : {(size_t)1 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)512 << 20}) {
    const unsigned entry_mask = (unsigned)(bytes / 8) - 1;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(gather_k, dim3(blocks), dim3(256), 0, 0, (const float2*)table, entry_mask, iters, table);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather_k, dim3(blocks), dim3(256), 0, 0, (const float2*)table, entry_mask, iters, table);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 3;
    printf("random float2 gathers, %4zu MB table   %8.3f ms  %7.2f G gathers/s\n", bytes >> 20, ms, (double)blocks * 256 * iters / ms * 1e-6);
  }
  return 0;
}

// What does one wave-level gather instruction cost a CU's texture-address / L1 path on gfx950, as a function of how its 64 lane
// addresses coalesce, of the load width and of where the table lives?  (The render kernels issue 128 per-lane-addressed loads
// per field sample; with fp16 MFMA products they ARE the kernel -- DESIGN.md section 4.12.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_rate_microbench.hip -o gather_rate && ./gather_rate
//
// Each wave issues UNROLL independent loads per iteration (addresses from a per-lane LCG, shaped by the pattern), sums them,
// and loops; 8 waves per CU (2 per SIMD), every CU busy.  Reported: cycles of CU time per wave-level load instruction
// (kernel time x clock / instructions per CU) and lanes per clock per CU.
//   pattern  lines touched by one instruction
//   same     1   (all 64 lanes inside one 128-byte line ... two lines for 8-byte loads)
//   row16    4   (each 16-lane row inside one line: consecutive samples of a ray in one coarse cell)
//   quad     16  (each quad inside one line)
//   pair     32  (lane pairs share a line: the x-neighbours of a dense level)
//   lane     64  (every lane its own random line: a fine hashed level)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

constexpr int UNROLL = 16;

// GROUP: lanes per shared line (64 same, 16 row16, 4 quad, 2 pair, 1 lane).  W: dwords per load (1, 2, 4).
// POLICY: 0 plain loads, 1 agent-scope relaxed atomic loads (global_load ... sc1: served by the L2, no L1 allocation),
// 2 non-temporal loads (nt).
template <int GROUP, int W, int POLICY = 0>
__global__ void __launch_bounds__(512) gather(const unsigned* __restrict__ table, unsigned line_mask, int iters, unsigned* out) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned grp = lane / GROUP, within = lane % GROUP;
  unsigned h = mix((wave * 64u + grp) * 2654435761u + 12345u);
  unsigned acc = 0;
  // a 128-byte line holds 32 dwords; lane `within` of a group reads dword(s) at (within * W) % 32 of its group's line
  const unsigned dw = (within * W) & 31u;
  for (int it = 0; it < iters; ++it) {
    unsigned v[UNROLL][W];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      h = h * 1664525u + 1013904223u;  // one v_mad per address; the high bits of an LCG are the good ones
      const unsigned line = (h >> 7) & line_mask;
      const unsigned* p = table + (size_t)line * 32u + dw;
      if (W == 1) {
        if (POLICY == 0) v[u][0] = *p;
        if (POLICY == 1) v[u][0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (POLICY == 2) v[u][0] = __builtin_nontemporal_load(p);
      }
      if (W == 2) {
        unsigned long long t = 0;
        const unsigned long long* p8 = reinterpret_cast<const unsigned long long*>(p);
        if (POLICY == 0) t = *p8;
        if (POLICY == 1) t = __hip_atomic_load(p8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (POLICY == 2) t = __builtin_nontemporal_load(p8);
        v[u][0] = (unsigned)t;
        v[u][1] = (unsigned)(t >> 32);
      }
      if (W == 4) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[u][0] = t.x;
        v[u][1] = t.y;
        v[u][2] = t.z;
        v[u][3] = t.w;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) acc += v[u][w];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int GROUP, int W, int POLICY = 0>
static float run(const unsigned* table, unsigned line_mask, int iters, unsigned* out, int blocks) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL((gather<GROUP, W, POLICY>), dim3(blocks), dim3(512), 0, 0, table, line_mask, iters / 4, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL((gather<GROUP, W, POLICY>), dim3(blocks), dim3(512), 0, 0, table, line_mask, iters, out);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const size_t max_bytes = 256u << 20;
  unsigned* table;
  unsigned* out;
  (void)hipMalloc(&table, max_bytes);
  (void)hipMemset(table, 1, max_bytes);
  const int blocks = 256;  // one 512-thread workgroup per CU: 8 waves, 2 per SIMD
  (void)hipMalloc(&out, (size_t)blocks * 512 * sizeof(unsigned));
  const double clock_ghz = 2.4;
  struct Size { const char* name; size_t bytes; int iters; };
  const Size sizes[] = {{"16 KB (L1)", 16u << 10, 400}, {"2 MB (L2)", 2u << 20, 200}, {"25 MB (MALL, a tcnn fp16 table)", 25u << 20, 100},
                        {"64 MB (MALL, the fp32 table)", 64u << 20, 100}};
  printf("%-34s %-6s %-7s %10s %14s %12s\n", "table", "width", "pattern", "ms", "cyc/instr/CU", "lanes/clk/CU");
  for (const Size& sz : sizes) {
    // line_mask must be 2^k - 1: round the line count down to a power of two
    unsigned lines = (unsigned)(sz.bytes / 128);
    unsigned p2 = 1;
    while (p2 * 2 <= lines) p2 *= 2;
    const unsigned mask = p2 - 1;
    auto report = [&](const char* w, const char* pat, float ms) {
      const double instr_per_cu = 8.0 * sz.iters * UNROLL;
      const double cyc = ms * 1e-3 * clock_ghz * 1e9 / instr_per_cu;
      printf("%-34s %-6s %-7s %10.3f %14.1f %12.2f\n", sz.name, w, pat, ms, cyc, 64.0 / cyc);
    };
#define ROW(W, WN)                                                            \
  report(WN, "same", run<64, W>(table, mask, sz.iters, out, blocks));         \
  report(WN, "row16", run<16, W>(table, mask, sz.iters, out, blocks));        \
  report(WN, "quad", run<4, W>(table, mask, sz.iters, out, blocks));          \
  report(WN, "pair", run<2, W>(table, mask, sz.iters, out, blocks));          \
  report(WN, "lane", run<1, W>(table, mask, sz.iters, out, blocks));
    ROW(1, "4 B")
    ROW(2, "8 B")
    ROW(4, "16 B")
#undef ROW
    // cache policies on the patterns that matter for a fine hashed level (every lane / every lane pair its own line)
    report("8 B", "lane sc1", run<1, 2, 1>(table, mask, sz.iters, out, blocks));
    report("8 B", "pair sc1", run<2, 2, 1>(table, mask, sz.iters, out, blocks));
    report("8 B", "lane nt", run<1, 2, 2>(table, mask, sz.iters, out, blocks));
    report("8 B", "pair nt", run<2, 2, 2>(table, mask, sz.iters, out, blocks));
    report("4 B", "lane sc1", run<1, 1, 1>(table, mask, sz.iters, out, blocks));
    report("4 B", "lane nt", run<1, 1, 2>(table, mask, sz.iters, out, blocks));
  }
  return 0;
}

// Can the fine-level gradient scatter of the training backward leave the float-atomic path?  (the field
// backward issues 127e6 float-atomic requests for its nine fine levels at 65 536 rays.)  Alternative priced here: every
// workgroup APPENDS its (entry, g0, g1) records to private per-bucket lists (bucket = a slice of one level's table; slot numbers
// from an LDS counter; plain 12-byte stores), and a second kernel -- one workgroup per bucket -- sums a bucket's lists in LDS and
// adds the slice to the table with plain loads and stores.
//
//   hipcc -O3 --offload-arch=gfx950 [-DSLICES_PER_LEVEL=28|56] [-DFOLD_MODE=m] tools/bucket_scatter_microbench.hip -o bucket_scatter
//
// Workload: LEVELS x 2^19 entries x 2 floats, UPDATES random (level, entry) records per launch (uniform: hashed levels).
// FOLD_MODE: 0 two ds_add_f32 per record; 1 list reads only; 2 ds_add_f32 only; 3 one ds_add_u64; 4 two ds_add_f64 (needs 56
// slices: 16 B of LDS per entry); 5 two ds_add_u64 (64-bit fixed point).  Results (MI355X, profiles/r03_append_scatter.txt), 226e6
// records: naive atomics 21.7 ms (the kernels' x-edge form needs ~6); append 1.05 ms (252 lists per workgroup) / 1.72 ms (504);
// fold 0.65 ms reading only, 2.28 ms with ds_add_f32 -- a third of a lane per clock and CU, whatever the unroll --, 0.73 ms with
// ds_add_f64 or ds_add_u64, which run at the rate the records stream in.  Built into field_backward_mfma_kernel (doubles, 56 slices,
// per-level list capacities, overflow to the atomics; gradients equal to 2e-6) it did NOT pay: the kernel stayed at 10.2 ms, the
// fold added 0.73 ms, the iteration went 19.4 -> 20.1 ms.  The fine levels' atomics were already hidden under the kernel's other
// phases -- switching their scatter off altogether only buys 1.3 ms.  The integration was removed again; this file is the record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef FOLD_MODE
#define FOLD_MODE 0
#endif
#ifndef SLICES_PER_LEVEL
#define SLICES_PER_LEVEL 28  // 9 x 28 = 252 buckets <= 256 CUs, 18 725 entries = 149.8 KB of LDS each (56: two workgroups per CU)
#endif
constexpr int LEVELS = 9, LOG2_T = 19;
constexpr int SLICES = SLICES_PER_LEVEL;
constexpr int SLICE_LEN = ((1 << LOG2_T) + SLICES - 1) / SLICES;
constexpr int BUCKETS = LEVELS * SLICES;
constexpr int ACC_BYTES = (FOLD_MODE == 4 || FOLD_MODE == 5 ? 16 : 8) * SLICE_LEN;
constexpr int PRODUCERS = 256, PTHREADS = 512;
#ifndef FOLD_MODE
#define FOLD_MODE 0
#endif
#ifndef FOLD_UNROLL
#define FOLD_UNROLL 4
#endif

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

struct Rec { unsigned idx; float g0, g1; };

// baseline: two float atomics per record on the table
__global__ void __launch_bounds__(PTHREADS) scatter_atomic(float* table, long long per_thread) {
  const unsigned tid = blockIdx.x * PTHREADS + threadIdx.x;
  unsigned h = mix(tid * 2654435761u + 99u);
  for (long long i = 0; i < per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned r = mix(h);
    const unsigned level = r % LEVELS, e = (r >> 8) & ((1u << LOG2_T) - 1);
    float* p = table + ((size_t)level << (LOG2_T + 1)) + 2 * e;
    atomicAdd(p, 1.0f);
    atomicAdd(p + 1, 0.5f);
  }
}

// producer: append to the workgroup's private list of the record's bucket; lists that are full fall back to the atomics
__global__ void __launch_bounds__(PTHREADS) scatter_append(Rec* lists, unsigned* counts, int cap, float* table, long long per_thread) {
  __shared__ unsigned cnt[BUCKETS];
  for (int i = threadIdx.x; i < BUCKETS; i += PTHREADS) cnt[i] = 0;
  __syncthreads();
  const unsigned tid = blockIdx.x * PTHREADS + threadIdx.x;
  unsigned h = mix(tid * 2654435761u + 99u);
  Rec* mine = lists + (size_t)blockIdx.x * BUCKETS * cap;
  for (long long i = 0; i < per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned r = mix(h);
    const unsigned level = r % LEVELS, e = (r >> 8) & ((1u << LOG2_T) - 1);
    const unsigned sl = e / SLICE_LEN;
    const unsigned b = level * SLICES + sl;
    const unsigned slot = atomicAdd(&cnt[b], 1u);
    if (slot < (unsigned)cap) {
      Rec rec = {e - sl * SLICE_LEN, 1.0f, 0.5f};
      mine[(size_t)b * cap + slot] = rec;
    } else {
      float* p = table + ((size_t)level << (LOG2_T + 1)) + 2 * e;
      atomicAdd(p, 1.0f);
      atomicAdd(p + 1, 0.5f);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BUCKETS; i += PTHREADS) counts[blockIdx.x * BUCKETS + i] = min(cnt[i], (unsigned)cap);
}

// consumer: one workgroup per bucket
__global__ void __launch_bounds__(1024) fold_buckets(const Rec* lists, const unsigned* counts, int cap, float* table) {
  extern __shared__ float acc[];  // 2 * SLICE_LEN floats
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < ACC_BYTES / 4; i += 1024) acc[i] = 0.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float keep = 0.f;
  for (int p = wave; p < PRODUCERS; p += 16) {
    const unsigned n = counts[p * BUCKETS + b];
    const Rec* src = lists + ((size_t)p * BUCKETS + b) * cap;
    for (unsigned i0 = 0; i0 < n; i0 += 64 * FOLD_UNROLL) {
      Rec r[FOLD_UNROLL];
#pragma unroll
      for (int u = 0; u < FOLD_UNROLL; ++u) {
        const unsigned i = i0 + 64 * u + lane;
#if FOLD_MODE == 2  // LDS atomics only (timing): no list reads
        r[u] = Rec{mix(i * 747796405u + p) % SLICE_LEN, 1.f, 0.5f};
#else
        r[u] = i < n ? src[i] : Rec{0u, 0.f, 0.f};
#endif
      }
#pragma unroll
      for (int u = 0; u < FOLD_UNROLL; ++u) {
        if (i0 + 64 * u + lane < n) {
#if FOLD_MODE == 1  // loads only (timing)
          keep += r[u].g0 + r[u].g1 + (float)r[u].idx;
#elif FOLD_MODE == 3  // one 8-byte LDS atomic per record (timing: u64 integer add on the float pair's bits)
          atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * r[u].idx]), (unsigned long long)__float_as_uint(r[u].g0));
#elif FOLD_MODE == 4  // (timing, build with SLICES_PER_LEVEL=56) two f64 LDS atomics per record
          atomicAdd(reinterpret_cast<double*>(acc) + 2 * r[u].idx, (double)r[u].g0);
          atomicAdd(reinterpret_cast<double*>(acc) + 2 * r[u].idx + 1, (double)r[u].g1);
#elif FOLD_MODE == 5  // (timing, SLICES_PER_LEVEL=56) two u64 LDS atomics per record: 64-bit fixed point
          atomicAdd(reinterpret_cast<unsigned long long*>(acc) + 2 * r[u].idx, (unsigned long long)(long long)(r[u].g0 * 0x1p40f));
          atomicAdd(reinterpret_cast<unsigned long long*>(acc) + 2 * r[u].idx + 1, (unsigned long long)(long long)(r[u].g1 * 0x1p40f));
#else
          atomicAdd(&acc[2 * r[u].idx], r[u].g0);
          atomicAdd(&acc[2 * r[u].idx + 1], r[u].g1);
#endif
        }
      }
    }
  }
  if (keep == 12345.678f) acc[0] = keep;
  __syncthreads();
  const int level = b / SLICES, slice = b % SLICES;
  const int len = min(SLICE_LEN, (1 << LOG2_T) - slice * SLICE_LEN);
  float* dst = table + ((size_t)level << (LOG2_T + 1)) + 2 * (size_t)slice * SLICE_LEN;
  for (int i = threadIdx.x; i < 2 * len; i += 1024) dst[i] += acc[i];  // += : the fallback atomics may have written
}

int main() {
  const size_t table_floats = (size_t)LEVELS << (LOG2_T + 1);
  float* table;
  (void)hipMalloc(&table, table_floats * 4);
  unsigned* counts;
  (void)hipMalloc(&counts, (size_t)PRODUCERS * BUCKETS * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fold_buckets), hipFuncAttributeMaxDynamicSharedMemorySize, ACC_BYTES);
  hipEvent_t e0, e1, e2;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
  printf("%12s %10s %12s %12s %12s %12s %10s\n", "updates", "cap", "atomic ms", "append ms", "fold ms", "sum ms", "check");
  for (long long per_thread : {216LL, 1728LL}) {  // 28.3e6 (4 096 rays x 48 x 16... scaled) and 226e6 records (65 536 rays: 3.1e6 samples x 8 corners x 9 levels)
    const long long updates = per_thread * PRODUCERS * PTHREADS;
    const int cap = (int)(updates / PRODUCERS / BUCKETS * 5 / 4) + 64;
    Rec* lists;
    if (hipMalloc(&lists, (size_t)PRODUCERS * BUCKETS * cap * sizeof(Rec)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float ms_a = 0, ms_p = 0, ms_f = 0;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemset(table, 0, table_floats * 4);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(scatter_atomic, dim3(PRODUCERS), dim3(PTHREADS), 0, 0, table, per_thread);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms_a, e0, e1);
    }
    std::vector<float> ref(table_floats);
    (void)hipMemcpy(ref.data(), table, table_floats * 4, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemset(table, 0, table_floats * 4);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(scatter_append, dim3(PRODUCERS), dim3(PTHREADS), 0, 0, lists, counts, cap, table, per_thread);
      (void)hipEventRecord(e1, 0);
      hipLaunchKernelGGL(fold_buckets, dim3(BUCKETS), dim3(1024), ACC_BYTES, 0, lists, counts, cap, table);
      (void)hipEventRecord(e2, 0);
      (void)hipEventSynchronize(e2);
      (void)hipEventElapsedTime(&ms_p, e0, e1);
      (void)hipEventElapsedTime(&ms_f, e1, e2);
    }
    std::vector<float> got(table_floats);
    (void)hipMemcpy(got.data(), table, table_floats * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < table_floats; ++i) bad += got[i] != ref[i];  // sums of 1.0 / 0.5: exact in any order
    printf("%12lld %10d %12.3f %12.3f %12.3f %12.3f %10zu\n", updates, cap, ms_a, ms_p, ms_f, ms_p + ms_f, bad);
    (void)hipFree(lists);
  }
  return 0;
}

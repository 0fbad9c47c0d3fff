// Exporter compaction kernels: threshold masks + stream compaction on device, so the exporters move only the kept
// points over PCIe (the reference boolean-indexes and .cpu()s every batch: fruit_nerf/export/exporter_utils.py:127-153,
// fruit_nerf/export/exporter_utils_nerfacto.py:156-183).  Wave-aggregated appends: one returning atomic per wave and set.
#include "composite_dev.hpp"

namespace cn {

__device__ __forceinline__ long long wave_append(bool keep, unsigned long long* counter) {
  unsigned long long m = __ballot(keep);
  if (m == 0) return -1;
  int lane = lane_id();
  int leader = __ffsll((long long)m) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
  base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)(base & 0xffffffffu), leader);
  unsigned long long below = m & ((1ull << lane) - 1ull);
  return keep ? (long long)(base + __popcll(below)) : -1;
}

struct ExportSets {
  float* points[3];
  float* colors[3];
};

__global__ void __launch_bounds__(256)
export_compact_kernel(const float* __restrict__ pos, const float* __restrict__ rgb, const float* __restrict__ sem,
                      const float* __restrict__ den, long long n, float sem_thresh, float den_thresh,
                      long long capacity, ExportSets sets, unsigned long long* __restrict__ counts) {
  const long long nround = (n + 63) / 64 * 64;  // whole waves stay converged for the ballots
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nround;
       i += (long long)gridDim.x * blockDim.x) {
    bool in = i < n;
    float s = in ? sem[i] : 0.f, d = in ? den[i] : 0.f;
    bool m_den = in && d >= den_thresh;
    bool m_sem = in && s >= sem_thresh;
    bool m_lab = in && semantics_label(s) >= 0.999f;
    bool keep[3] = {m_lab && m_den, m_sem && m_den, m_den};
    float fourth[3] = {sigmoidf(s), sigmoidf(s), sigmoidf(d)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      long long dst = wave_append(keep[k], counts + k);
      if (dst >= 0 && dst < capacity) {
        float* p = sets.points[k] + 3 * dst;
        float* c = sets.colors[k] + 4 * dst;
        p[0] = pos[3 * i];
        p[1] = pos[3 * i + 1];
        p[2] = pos[3 * i + 2];
        c[0] = rgb[3 * i];
        c[1] = rgb[3 * i + 1];
        c[2] = rgb[3 * i + 2];
        c[3] = fourth[k];
      }
    }
  }
}

__global__ void __launch_bounds__(256)
pointcloud_compact_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ depth,
                          const float* __restrict__ rgb, const float* __restrict__ cmap, long long n,
                          long long capacity, float* __restrict__ points, float* __restrict__ colors,
                          float* __restrict__ dirs, unsigned long long* __restrict__ count) {
  const long long nround = (n + 63) / 64 * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nround;
       i += (long long)gridDim.x * blockDim.x) {
    bool keep = i < n && cmap[3 * i] > 0.f;
    long long dst = wave_append(keep, count);
    if (dst >= 0 && dst < capacity) {
      float t = depth[i];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        points[3 * dst + k] = o[3 * i + k] + d[3 * i + k] * t;
        colors[3 * dst + k] = rgb[3 * i + k];
        if (dirs) dirs[3 * dst + k] = d[3 * i + k];
      }
    }
  }
}

}  // namespace cn

extern "C" int cn_export_compact(const float* positions, const float* rgb, const float* semantics,
                                 const float* density, int64_t num_samples, float sem_thresh, float den_thresh,
                                 int64_t capacity, float* const* points3, float* const* colors3, int64_t* counts,
                                 cn_stream_t stream) {
  CN_REQUIRE(positions && rgb && semantics && density && points3 && colors3 && counts, CN_ERR_INVALID,
             "cn_export_compact: null argument");
  CN_REQUIRE(capacity >= 0, CN_ERR_INVALID, "cn_export_compact: negative capacity");
  if (num_samples <= 0) return CN_OK;
  cn::ExportSets sets;
  for (int k = 0; k < 3; ++k) {
    CN_REQUIRE(points3[k] && colors3[k], CN_ERR_INVALID, "cn_export_compact: null output set %d", k);
    sets.points[k] = points3[k];
    sets.colors[k] = colors3[k];
  }
  hipLaunchKernelGGL(cn::export_compact_kernel, dim3(cn::grid_for(num_samples, 256, 4096)), dim3(256), 0,
                     cn::as_stream(stream), positions, rgb, semantics, density, (long long)num_samples, sem_thresh,
                     den_thresh, (long long)capacity, sets, reinterpret_cast<unsigned long long*>(counts));
  return cn::check_launch("cn_export_compact");
}

extern "C" int cn_pointcloud_compact(const float* origins, const float* directions, const float* depth,
                                     const float* rgb, const float* semantics_colormap, int64_t num_rays,
                                     int64_t capacity, float* points, float* colors, float* view_dirs, int64_t* count,
                                     cn_stream_t stream) {
  CN_REQUIRE(origins && directions && depth && rgb && semantics_colormap && points && colors && count, CN_ERR_INVALID,
             "cn_pointcloud_compact: null argument");
  if (num_rays <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::pointcloud_compact_kernel, dim3(cn::grid_for(num_rays, 256, 4096)), dim3(256), 0,
                     cn::as_stream(stream), origins, directions, depth, rgb, semantics_colormap, (long long)num_rays,
                     (long long)capacity, points, colors, view_dirs, reinterpret_cast<unsigned long long*>(count));
  return cn::check_launch("cn_pointcloud_compact");
}

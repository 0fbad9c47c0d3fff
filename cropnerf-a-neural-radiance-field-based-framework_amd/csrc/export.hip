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

// ---- generate_point_cloud with several of the reference's calls per launch -------------------------------------------------
// The reference draws `rays_per_call` random pixels, renders them, appends the kept points and checks the running count, call
// after call (exporter_utils_nerfacto.py:125-183; 2 048 rays per call at debug/exporter_nerfacto.py:91, 32 768 for upstream
// ns-export): the cloud is the union of ALL points of calls 0 .. c*, c* = the first call at which the count reaches the
// target.  One launch over K calls' rays gives the same cloud if (1) the pixel draws of call c do not depend on how calls are
// grouped -- pixel_sample_kernel: a counter-based stream, value = hash(seed, call, ray, component) -- and (2) the append stops
// at the same call boundary: per-call kept counts -> running count -> ray limit (c* + 1) * rays_per_call, on the device.

__device__ __forceinline__ unsigned pixel_bits24(unsigned long long seed, unsigned long long call, unsigned ray, unsigned comp) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (3ull * call + comp + 1ull);
  z ^= (unsigned long long)ray * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;  // splitmix64 finaliser
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 40);
}

__global__ void __launch_bounds__(256)
pixel_sample_kernel(unsigned long long seed, const long long* __restrict__ first_call, int num_calls, int rays_per_call,
                    int num_cameras, int height, int width, int64_t* __restrict__ ray_indices) {
  const long long total = (long long)num_calls * rays_per_call;
  const long long call0 = *first_call;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long call = (unsigned long long)(call0 + i / rays_per_call);
    const unsigned ray = (unsigned)(i % rays_per_call);
    // floor(u * n) for u = bits / 2^24 (PixelSampler: floor(rand((B, 3)) * [num_images, H, W])), in integers
    ray_indices[3 * i + 0] = (int64_t)(((unsigned long long)pixel_bits24(seed, call, ray, 0) * (unsigned)num_cameras) >> 24);
    ray_indices[3 * i + 1] = (int64_t)(((unsigned long long)pixel_bits24(seed, call, ray, 1) * (unsigned)height) >> 24);
    ray_indices[3 * i + 2] = (int64_t)(((unsigned long long)pixel_bits24(seed, call, ray, 2) * (unsigned)width) >> 24);
  }
}

__global__ void __launch_bounds__(256)
pointcloud_call_count_kernel(const float* __restrict__ cmap, long long n, long long rays_per_call,
                             unsigned long long* __restrict__ call_counts) {
  const long long nround = (n + 63) / 64 * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nround; i += (long long)gridDim.x * blockDim.x) {
    const bool keep = i < n && cmap[3 * i] > 0.f;
    const long long call = i < n ? i / rays_per_call : -1;
    const long long call0 = ((long long)__builtin_amdgcn_readfirstlane((int)(call >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(call & 0xffffffff));
    const unsigned long long m = __ballot(keep);
    if (__ballot(i < n && call != call0) == 0ull) {  // the wave lies inside one call (always, when 64 divides rays_per_call)
      if (m && lane_id() == __ffsll((long long)m) - 1) atomicAdd(call_counts + call0, (unsigned long long)__popcll(m));
    } else if (keep) {
      atomicAdd(call_counts + call, 1ull);
    }
  }
}

__global__ void pointcloud_call_limit_kernel(const unsigned long long* __restrict__ call_counts, long long num_calls,
                                             long long rays_per_call, long long n, long long target,
                                             const unsigned long long* __restrict__ count, long long* __restrict__ ray_limit) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long cum = *count;
  long long limit = n;
  if ((long long)cum >= target) {
    limit = 0;  // an earlier call reached the target: the reference's loop has ended
  } else {
    for (long long c = 0; c < num_calls; ++c) {
      cum += call_counts[c];
      if ((long long)cum >= target) {
        limit = (c + 1) * rays_per_call < n ? (c + 1) * rays_per_call : n;
        break;
      }
    }
  }
  *ray_limit = limit;
}

__global__ void __launch_bounds__(256)
pointcloud_compact_limited_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ depth,
                                  const float* __restrict__ rgb, const float* __restrict__ cmap,
                                  const long long* __restrict__ ray_limit, long long capacity, float* __restrict__ points,
                                  float* __restrict__ colors, float* __restrict__ dirs, unsigned long long* __restrict__ count) {
  const long long n = *ray_limit;
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

extern "C" int cn_pixel_sample(uint64_t seed, const int64_t* first_call, int32_t num_calls, int32_t rays_per_call,
                               int32_t num_cameras, int32_t height, int32_t width, int64_t* ray_indices,
                               cn_stream_t stream) {
  CN_REQUIRE(first_call && ray_indices, CN_ERR_INVALID, "cn_pixel_sample: null argument");
  CN_REQUIRE(num_calls >= 0 && rays_per_call > 0, CN_ERR_INVALID, "cn_pixel_sample: %d calls of %d rays", num_calls,
             rays_per_call);
  CN_REQUIRE(num_cameras > 0 && height > 0 && width > 0 && num_cameras < (1 << 24) && height < (1 << 24) && width < (1 << 24),
             CN_ERR_INVALID, "cn_pixel_sample: %d cameras of %d x %d", num_cameras, height, width);
  if (num_calls == 0) return CN_OK;
  hipLaunchKernelGGL(cn::pixel_sample_kernel, dim3(cn::grid_for((long long)num_calls * rays_per_call, 256, 4096)), dim3(256), 0,
                     cn::as_stream(stream), (unsigned long long)seed, reinterpret_cast<const long long*>(first_call), num_calls,
                     rays_per_call, num_cameras, height, width, ray_indices);
  return cn::check_launch("cn_pixel_sample");
}

extern "C" int cn_pointcloud_compact_calls(const float* origins, const float* directions, const float* depth, const float* rgb,
                                           const float* semantics_colormap, int64_t num_rays, int64_t rays_per_call,
                                           int64_t target_points, int64_t capacity, float* points, float* colors,
                                           float* view_dirs, int64_t* count, int64_t* call_counts, int64_t* ray_limit,
                                           cn_stream_t stream) {
  CN_REQUIRE(origins && directions && depth && rgb && semantics_colormap && points && colors && count && call_counts &&
                 ray_limit, CN_ERR_INVALID, "cn_pointcloud_compact_calls: null argument");
  CN_REQUIRE(rays_per_call > 0 && num_rays >= 0, CN_ERR_INVALID, "cn_pointcloud_compact_calls: %lld rays in calls of %lld",
             (long long)num_rays, (long long)rays_per_call);
  if (num_rays == 0) return CN_OK;
  const long long calls = (num_rays + rays_per_call - 1) / rays_per_call;
  hipStream_t s = cn::as_stream(stream);
  hipError_t e = hipMemsetAsync(call_counts, 0, sizeof(int64_t) * (size_t)calls, s);
  CN_REQUIRE(e == hipSuccess, CN_ERR_LAUNCH, "cn_pointcloud_compact_calls: %s", hipGetErrorString(e));
  const unsigned grid = cn::grid_for(num_rays, 256, 4096);
  hipLaunchKernelGGL(cn::pointcloud_call_count_kernel, dim3(grid), dim3(256), 0, s, semantics_colormap, (long long)num_rays,
                     (long long)rays_per_call, reinterpret_cast<unsigned long long*>(call_counts));
  hipLaunchKernelGGL(cn::pointcloud_call_limit_kernel, dim3(1), dim3(64), 0, s,
                     reinterpret_cast<const unsigned long long*>(call_counts), calls, (long long)rays_per_call,
                     (long long)num_rays, (long long)target_points, reinterpret_cast<const unsigned long long*>(count),
                     reinterpret_cast<long long*>(ray_limit));
  hipLaunchKernelGGL(cn::pointcloud_compact_limited_kernel, dim3(grid), dim3(256), 0, s, origins, directions, depth, rgb,
                     semantics_colormap, reinterpret_cast<const long long*>(ray_limit), (long long)capacity, points, colors,
                     view_dirs, reinterpret_cast<unsigned long long*>(count));
  return cn::check_launch("cn_pointcloud_compact_calls");
}

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

// Super-cluster stage of the segmenter (segmentation/segmenter.py:69-86: get_super_clusters =
// voxel_down_sample -> cluster_dbscan(eps = 20 voxels, min_points = 30) -> drop noise -> remove_statistical_outlier
// (csrc/knn.hip)).  The caller bins the points on a uniform grid with cell size >= eps (sorted by cell, cell offsets);
// everything here is one thread per point over the 3x3x3 block of cells around it.
//
//   cn_segment_mean   open3d voxel_down_sample: the points (any number of channels) of one voxel are averaged
//   cn_dbscan         density-based clustering with DBSCAN's definitions: a point is a CORE point when at least
//                     min_points points (itself included) lie within eps; clusters are the connected components of the
//                     core points under "within eps" (lock-free union-find, the smaller index becomes the root); a
//                     non-core point within eps of a core point is a BORDER point of that point's cluster (where open3d /
//                     sklearn take the first cluster that reaches it in their scan order, this takes the core neighbour with
//                     the smallest original index -- DBSCAN leaves that choice open); the rest is noise (-1).
#include "cn_common.hpp"

namespace cn {

__global__ void __launch_bounds__(256)
segment_mean_kernel(const float* __restrict__ vals, const int* __restrict__ seg_start, long long nseg, int C,
                    float* __restrict__ out) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nseg * C; t += (long long)gridDim.x * blockDim.x) {
    const long long sgm = t / C;
    const int c = (int)(t - sgm * C);
    const int lo = seg_start[sgm], hi = seg_start[sgm + 1];
    double acc = 0.0;  // open3d accumulates in double
    for (int i = lo; i < hi; ++i) acc += (double)vals[(long long)i * C + c];
    out[t] = (float)(acc / (double)(hi - lo));
  }
}

struct GridArgs {
  const float* pts;
  const int* cell_start;
  int gx, gy, gz;
  float ox, oy, oz, inv_h;
  long long n;
};

template <typename F>
__device__ __forceinline__ void for_each_neighbour(const GridArgs& G, long long i, float eps2, F&& f) {
  const float px = G.pts[3 * i], py = G.pts[3 * i + 1], pz = G.pts[3 * i + 2];
  const int cx = min(max((int)floorf((px - G.ox) * G.inv_h), 0), G.gx - 1);
  const int cy = min(max((int)floorf((py - G.oy) * G.inv_h), 0), G.gy - 1);
  const int cz = min(max((int)floorf((pz - G.oz) * G.inv_h), 0), G.gz - 1);
  for (int z = max(cz - 1, 0); z <= min(cz + 1, G.gz - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, G.gy - 1); ++y)
      for (int x = max(cx - 1, 0); x <= min(cx + 1, G.gx - 1); ++x) {
        const long long c = ((long long)z * G.gy + y) * G.gx + x;
        const int lo = G.cell_start[c], hi = G.cell_start[c + 1];
        for (int q = lo; q < hi; ++q) {
          const float ex = G.pts[3 * q] - px, ey = G.pts[3 * q + 1] - py, ez = G.pts[3 * q + 2] - pz;
          if (ex * ex + ey * ey + ez * ez <= eps2) f(q);
        }
      }
}

__global__ void __launch_bounds__(256) dbscan_count_kernel(GridArgs G, float eps2, int* __restrict__ count, int* __restrict__ parent) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < G.n; i += (long long)gridDim.x * blockDim.x) {
    int c = 0;
    for_each_neighbour(G, i, eps2, [&](int) { ++c; });
    count[i] = c;
    parent[i] = (int)i;
  }
}

__device__ __forceinline__ int uf_find(int* parent, int x) {
  int p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    const int gp = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) atomicCAS(parent + x, p, gp);  // path halving
    x = p;
    p = gp;
  }
  return x;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }  // hook the larger root under the smaller
    if (atomicCAS(parent + a, a, b) == a) return;
  }
}

__global__ void __launch_bounds__(256)
dbscan_union_kernel(GridArgs G, float eps2, int min_points, const int* __restrict__ count, int* __restrict__ parent) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < G.n; i += (long long)gridDim.x * blockDim.x) {
    if (count[i] < min_points) continue;
    for_each_neighbour(G, i, eps2, [&](int q) {
      if (q < (int)i && count[q] >= min_points) uf_union(parent, (int)i, q);
    });
  }
}

// root (sorted index) of every point's cluster, -1 for noise
__global__ void __launch_bounds__(256)
dbscan_assign_kernel(GridArgs G, float eps2, int min_points, const int* __restrict__ count, int* __restrict__ parent,
                     const int64_t* __restrict__ order, int* __restrict__ root) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < G.n; i += (long long)gridDim.x * blockDim.x) {
    if (count[i] >= min_points) {
      root[i] = uf_find(parent, (int)i);
      continue;
    }
    long long best = -1;
    int best_q = -1;
    for_each_neighbour(G, i, eps2, [&](int q) {
      if (count[q] >= min_points && (best < 0 || order[q] < best)) {
        best = order[q];
        best_q = q;
      }
    });
    root[i] = best_q >= 0 ? uf_find(parent, best_q) : -1;
  }
}

}  // namespace cn

extern "C" int cn_segment_mean(const float* values_sorted, const int32_t* segment_start, int64_t num_segments,
                               int32_t channels, float* out, cn_stream_t stream) {
  CN_REQUIRE(channels >= 1, CN_ERR_INVALID, "cn_segment_mean: channels must be >= 1");
  if (num_segments <= 0) return CN_OK;
  CN_REQUIRE(values_sorted && segment_start && out, CN_ERR_INVALID, "cn_segment_mean: null argument");
  hipLaunchKernelGGL(cn::segment_mean_kernel, dim3(cn::grid_for(num_segments * channels, 256, 1 << 16)), dim3(256), 0,
                     cn::as_stream(stream), values_sorted, segment_start, (long long)num_segments, channels, out);
  return cn::check_launch("cn_segment_mean");
}

extern "C" int cn_dbscan(const float* points_sorted, const int32_t* cell_start, int32_t gx, int32_t gy, int32_t gz,
                         float origin_x, float origin_y, float origin_z, float cell_size, float eps, int32_t min_points,
                         const int64_t* order, int64_t num_points, int32_t* neighbour_count, int32_t* parent,
                         int32_t* root, cn_stream_t stream) {
  CN_REQUIRE(gx > 0 && gy > 0 && gz > 0 && cell_size > 0.f && eps > 0.f && min_points >= 1, CN_ERR_INVALID,
             "cn_dbscan: bad argument");
  CN_REQUIRE(cell_size >= eps, CN_ERR_INVALID, "cn_dbscan: the grid cells must be at least eps wide");
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(num_points < (1LL << 31), CN_ERR_INVALID, "cn_dbscan: at most 2^31-1 points");
  CN_REQUIRE(points_sorted && cell_start && order && neighbour_count && parent && root, CN_ERR_INVALID,
             "cn_dbscan: null argument");
  cn::GridArgs G{points_sorted, cell_start, gx, gy, gz, origin_x, origin_y, origin_z, 1.f / cell_size, (long long)num_points};
  const float eps2 = eps * eps;
  hipStream_t s = cn::as_stream(stream);
  const dim3 grid(cn::grid_for(num_points, 256, 1 << 16)), block(256);
  hipLaunchKernelGGL(cn::dbscan_count_kernel, grid, block, 0, s, G, eps2, neighbour_count, parent);
  hipLaunchKernelGGL(cn::dbscan_union_kernel, grid, block, 0, s, G, eps2, (int)min_points, neighbour_count, parent);
  hipLaunchKernelGGL(cn::dbscan_assign_kernel, grid, block, 0, s, G, eps2, (int)min_points, neighbour_count, parent, order,
                     root);
  return cn::check_launch("cn_dbscan");
}

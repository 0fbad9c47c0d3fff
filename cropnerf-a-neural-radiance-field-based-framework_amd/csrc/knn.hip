// Statistical outlier removal of the point-cloud exporter (the part that sits inside row a17):
// generate_point_cloud calls open3d's remove_statistical_outlier(nb_neighbors=20, std_ratio)
// (fruit_nerf/export/exporter_utils_nerfacto.py:194-199) on up to 10 M points.  open3d builds a KD-tree and, per
// point, averages the distances to its nb_neighbors nearest points (the point itself, at distance 0, included), then
// drops the points whose average exceeds cloud mean + std_ratio * cloud std.  The per-point k-nearest search is the cost;
// here it runs on a uniform grid: the caller bins the points (sorted by cell, cell start offsets), and one thread per
// point scans the 3x3x3 block of cells around its own, then further shells while the k-th best distance may still be
// beaten from outside the block.  The k best squared distances live in registers (sorted insertion, fully unrolled).
#include <algorithm>

#include "cn_common.hpp"

namespace cn {

constexpr int KNN_MAX_K = 32;

template <int K>
__global__ void __launch_bounds__(256)
knn_mean_distance_kernel(const float* __restrict__ pts /*sorted by cell*/, const int* __restrict__ cell_start,
                         int gx, int gy, int gz, float ox, float oy, float oz, float inv_h, float h, long long n,
                         int k_used, int max_ring, float* __restrict__ mean_out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int cx = min(max((int)floorf((px - ox) * inv_h), 0), gx - 1);
    const int cy = min(max((int)floorf((py - oy) * inv_h), 0), gy - 1);
    const int cz = min(max((int)floorf((pz - oz) * inv_h), 0), gz - 1);
    float best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = 3.0e38f;
    for (int ring = 0; ring <= max_ring; ++ring) {
      // the shell of cells at Chebyshev distance `ring` from (cx, cy, cz)
      for (int dz = -ring; dz <= ring; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= gz) continue;
        for (int dy = -ring; dy <= ring; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y >= gy) continue;
          const bool face = (dz == -ring || dz == ring || dy == -ring || dy == ring);
          const int step = face ? 1 : max(2 * ring, 1);  // interior rows of the shell: only the two end cells
          for (int dx = -ring; dx <= ring; dx += step) {
            const int x = cx + dx;
            if (x < 0 || x >= gx) continue;
            const long long c = ((long long)z * gy + y) * gx + x;
            const int lo = cell_start[c], hi = cell_start[c + 1];
            for (int q = lo; q < hi; ++q) {
              const float ex = pts[3 * q] - px, ey = pts[3 * q + 1] - py, ez = pts[3 * q + 2] - pz;
              float d = ex * ex + ey * ey + ez * ez;
              if (d < best[K - 1]) {
                // sorted insertion, fully unrolled
#pragma unroll
                for (int j = 0; j < K; ++j) {
                  const float lo_v = fminf(best[j], d);
                  d = fmaxf(best[j], d);
                  best[j] = lo_v;
                }
              }
            }
          }
        }
      }
      // everything outside the (2 ring + 1)^3 block is at least ring * h away (the point lies inside the centre cell)
      const float reach = (float)ring * h;
      if (best[k_used - 1 < K ? k_used - 1 : K - 1] <= reach * reach && ring >= 1) break;
    }
    float sum = 0.f;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (j < k_used && best[j] < 3.0e38f) {
        sum += sqrtf(best[j]);
        ++cnt;
      }
    }
    mean_out[i] = cnt > 0 ? sum / (float)cnt : -1.f;
  }
}

}  // namespace cn

extern "C" int cn_knn_mean_distance(const float* points_sorted, const int32_t* cell_start, int32_t gx, int32_t gy,
                                    int32_t gz, float origin_x, float origin_y, float origin_z, float cell_size,
                                    int64_t num_points, int32_t nb_neighbors, float* mean_distance,
                                    cn_stream_t stream) {
  CN_REQUIRE(gx > 0 && gy > 0 && gz > 0 && cell_size > 0.f, CN_ERR_INVALID, "cn_knn_mean_distance: bad grid");
  CN_REQUIRE(nb_neighbors >= 1 && nb_neighbors <= cn::KNN_MAX_K, CN_ERR_UNSUPPORTED,
             "cn_knn_mean_distance: nb_neighbors %d (max %d)", nb_neighbors, cn::KNN_MAX_K);
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(points_sorted && cell_start && mean_distance, CN_ERR_INVALID, "cn_knn_mean_distance: null argument");
  const int max_ring = std::max(gx, std::max(gy, gz));
  hipLaunchKernelGGL(cn::knn_mean_distance_kernel<cn::KNN_MAX_K>, dim3(cn::grid_for(num_points, 256, 1 << 16)), dim3(256),
                     0, cn::as_stream(stream), points_sorted, cell_start, gx, gy, gz, origin_x, origin_y, origin_z,
                     1.f / cell_size, cell_size, (long long)num_points, nb_neighbors, max_ring, mean_distance);
  return cn::check_launch("cn_knn_mean_distance");
}

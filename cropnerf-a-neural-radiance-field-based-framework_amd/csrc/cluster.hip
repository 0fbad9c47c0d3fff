// Super-cluster stage of the segmenter (segmentation/segmenter.py:69-86: get_super_clusters =
// voxel_down_sample -> cluster_dbscan(eps = 20 voxels, min_points = 30) -> drop noise -> remove_statistical_outlier
// (csrc/knn.hip)).  The caller bins the points (sorted by cell); everything here is one thread per point.
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

// ---------------------------------------------------------------------------------------------------------------
// DBSCAN on a sparse grid whose cells have a diagonal <= eps (cell size <= eps / sqrt 3): any two points of a cell are
// within eps of each other, and everything within eps of a point lies in the 5x5x5 block of cells around its own.
//   * a cell holding >= min_points points makes all of them core points of ONE cluster without a single distance;
//   * connectivity needs ONE core-core link per pair of neighbouring cells, not one per pair of points: a core point
//     links to its cell's representative, then looks into each neighbouring cell whose representative is not yet in its
//     component and stops at the first core point within eps.  With eps = 20 voxels (segmenter.py:76) a point has
//     thousands of neighbours; this does ~10^2 operations per point instead.
// The caller sorts the points by cell key ((z*dim_y + y)*dim_x + x, 64 bit) and passes the occupied cells only.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DB_ROWS = 25;            // (dy, dz) in [-2, 2]^2; the 5 cells of a row are consecutive keys
constexpr unsigned DB_FIRST_MASK = 0x1FFFFFFFu;  // row entry = first occupied cell | (number of occupied cells << 29)
constexpr int DB_NONE = 0x7FFFFFFF;

struct SparseGrid {
  const float* pts;              // [n,3] sorted by cell
  const long long* cell_keys;    // [m] ascending
  const int* cell_start;         // [m+1]
  const int* point_cell;         // [n]
  unsigned* rows;                // [m, 25]
  int* rep;                      // [m] smallest sorted index of a core point of the cell, DB_NONE if none
  long long dim_x, dim_y, dim_z;
  int m;
  long long n;
};

// per (occupied cell, row): the occupied cells of the row, found by binary search over the sorted keys
__global__ void __launch_bounds__(256) dbscan_rows_kernel(SparseGrid G) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < (long long)G.m * DB_ROWS;
       t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t / DB_ROWS), r = (int)(t - (long long)c * DB_ROWS);
    if (r == 0) G.rep[c] = DB_NONE;
    const long long key = G.cell_keys[c];
    const long long x = key % G.dim_x, yz = key / G.dim_x, y = yz % G.dim_y, z = yz / G.dim_y;
    const long long yy = y + (r % 5) - 2, zz = z + (r / 5) - 2;
    unsigned entry = 0;
    if (yy >= 0 && yy < G.dim_y && zz >= 0 && zz < G.dim_z) {
      const long long base = (zz * G.dim_y + yy) * G.dim_x;
      const long long klo = base + (x - 2 > 0 ? x - 2 : 0), khi = base + (x + 2 < G.dim_x - 1 ? x + 2 : G.dim_x - 1);
      int lo = 0, hi = G.m;  // lower bound of klo
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (G.cell_keys[mid] < klo) lo = mid + 1; else hi = mid;
      }
      unsigned cnt = 0;
      while (cnt < 5 && lo + (int)cnt < G.m && G.cell_keys[lo + cnt] <= khi) ++cnt;
      entry = (unsigned)lo | (cnt << 29);
    }
    G.rows[t] = entry;
  }
}

__device__ __forceinline__ bool within(const float* __restrict__ pts, int q, float px, float py, float pz, float eps2) {
  const float ex = pts[3 * q] - px, ey = pts[3 * q + 1] - py, ez = pts[3 * q + 2] - pz;
  return ex * ex + ey * ey + ez * ez <= eps2;
}

// neighbour_count saturates at min_points (all DBSCAN needs); core points announce themselves as cell representative
__global__ void __launch_bounds__(256)
dbscan_count_kernel(SparseGrid G, float eps2, int min_points, int* __restrict__ count, int* __restrict__ parent) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < G.n; i += (long long)gridDim.x * blockDim.x) {
    const int c = G.point_cell[i];
    int cnt = G.cell_start[c + 1] - G.cell_start[c];  // the whole cell is within eps
    if (cnt < min_points) {
      const float px = G.pts[3 * i], py = G.pts[3 * i + 1], pz = G.pts[3 * i + 2];
      for (int r = 0; r < DB_ROWS && cnt < min_points; ++r) {
        const unsigned e = G.rows[(long long)c * DB_ROWS + r];
        const int first = (int)(e & DB_FIRST_MASK), ncell = (int)(e >> 29);
        for (int cc = first; cc < first + ncell && cnt < min_points; ++cc) {
          if (cc == c) continue;
          const int hi = G.cell_start[cc + 1];
          for (int q = G.cell_start[cc]; q < hi && cnt < min_points; ++q) cnt += within(G.pts, q, px, py, pz, eps2) ? 1 : 0;
        }
      }
    }
    cnt = min(cnt, min_points);
    count[i] = cnt;
    parent[i] = (int)i;
    if (cnt >= min_points) atomicMin(G.rep + c, (int)i);
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

// every core point: link to the cell representative, then one link into each EARLIER neighbouring cell (the later ones
// look back at this one) unless that cell's representative is already in the same component
__global__ void __launch_bounds__(256)
dbscan_union_kernel(SparseGrid G, float eps2, int min_points, const int* __restrict__ count, int* __restrict__ parent) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < G.n; i += (long long)gridDim.x * blockDim.x) {
    if (count[i] < min_points) continue;
    const int c = G.point_cell[i];
    const int own = G.rep[c];
    if (own != (int)i) uf_union(parent, (int)i, own);
    const float px = G.pts[3 * i], py = G.pts[3 * i + 1], pz = G.pts[3 * i + 2];
    for (int r = 0; r <= DB_ROWS / 2; ++r) {  // rows of smaller keys, and the own row up to the own cell
      const unsigned e = G.rows[(long long)c * DB_ROWS + r];
      const int first = (int)(e & DB_FIRST_MASK), ncell = (int)(e >> 29);
      for (int cc = first; cc < first + ncell && cc < c; ++cc) {
        const int rq = G.rep[cc];
        if (rq == DB_NONE) continue;
        if (uf_find(parent, rq) == uf_find(parent, (int)i)) continue;
        const int hi = G.cell_start[cc + 1];
        for (int q = G.cell_start[cc]; q < hi; ++q)
          if (count[q] >= min_points && within(G.pts, q, px, py, pz, eps2)) {
            uf_union(parent, (int)i, q);
            break;
          }
      }
    }
  }
}

// root (sorted index) of every point's cluster, -1 for noise
__global__ void __launch_bounds__(256)
dbscan_assign_kernel(SparseGrid G, float eps2, int min_points, const int* __restrict__ count, int* __restrict__ parent,
                     const int64_t* __restrict__ order, int* __restrict__ root) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < G.n; i += (long long)gridDim.x * blockDim.x) {
    if (count[i] >= min_points) {
      root[i] = uf_find(parent, (int)i);
      continue;
    }
    const int c = G.point_cell[i];
    const float px = G.pts[3 * i], py = G.pts[3 * i + 1], pz = G.pts[3 * i + 2];
    long long best = -1;
    int best_q = -1;
    for (int r = 0; r < DB_ROWS; ++r) {
      const unsigned e = G.rows[(long long)c * DB_ROWS + r];
      const int first = (int)(e & DB_FIRST_MASK), ncell = (int)(e >> 29);
      for (int cc = first; cc < first + ncell; ++cc) {
        if (G.rep[cc] == DB_NONE) continue;  // no core point there
        const int hi = G.cell_start[cc + 1];
        for (int q = G.cell_start[cc]; q < hi; ++q)
          if (count[q] >= min_points && (best < 0 || order[q] < best) && within(G.pts, q, px, py, pz, eps2)) {
            best = order[q];
            best_q = q;
          }
      }
    }
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

extern "C" size_t cn_dbscan_workspace_bytes(int64_t num_cells) {
  return num_cells <= 0 ? 0 : (size_t)num_cells * (cn::DB_ROWS + 1) * sizeof(int32_t);
}

extern "C" int cn_dbscan(const float* points_sorted, const int64_t* cell_keys, const int32_t* cell_start,
                         const int32_t* point_cell, int64_t num_cells, int64_t dim_x, int64_t dim_y, int64_t dim_z,
                         float cell_size, float eps, int32_t min_points, const int64_t* order, int64_t num_points,
                         int32_t* neighbour_count, int32_t* parent, int32_t* root, void* workspace, size_t workspace_bytes,
                         cn_stream_t stream) {
  CN_REQUIRE(cell_size > 0.f && eps > 0.f && min_points >= 1, CN_ERR_INVALID, "cn_dbscan: bad argument");
  CN_REQUIRE(cell_size * 1.7320508f <= eps, CN_ERR_INVALID, "cn_dbscan: the diagonal of a grid cell must not exceed eps");
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(num_points < (1LL << 31), CN_ERR_INVALID, "cn_dbscan: at most 2^31-1 points");
  CN_REQUIRE(num_cells >= 1 && num_cells <= num_points && num_cells < (1LL << 29), CN_ERR_INVALID, "cn_dbscan: bad cell count");
  CN_REQUIRE(dim_x >= 1 && dim_y >= 1 && dim_z >= 1 && dim_x < (1LL << 20) && dim_y < (1LL << 20) && dim_z < (1LL << 20),
             CN_ERR_INVALID, "cn_dbscan: grid dimensions must be in [1, 2^20)");
  CN_REQUIRE(points_sorted && cell_keys && cell_start && point_cell && order && neighbour_count && parent && root && workspace,
             CN_ERR_INVALID, "cn_dbscan: null argument");
  CN_REQUIRE(workspace_bytes >= cn_dbscan_workspace_bytes(num_cells), CN_ERR_WORKSPACE, "cn_dbscan: workspace too small");
  unsigned* rows = reinterpret_cast<unsigned*>(workspace);
  int* rep = reinterpret_cast<int*>(rows + (size_t)num_cells * cn::DB_ROWS);
  cn::SparseGrid G{points_sorted, reinterpret_cast<const long long*>(cell_keys), cell_start, point_cell, rows, rep,
                   (long long)dim_x, (long long)dim_y, (long long)dim_z, (int)num_cells, (long long)num_points};
  const float eps2 = eps * eps;
  hipStream_t s = cn::as_stream(stream);
  const dim3 grid(cn::grid_for(num_points, 256, 1 << 16)), block(256);
  hipLaunchKernelGGL(cn::dbscan_rows_kernel, dim3(cn::grid_for(num_cells * cn::DB_ROWS, 256, 1 << 16)), block, 0, s, G);
  hipLaunchKernelGGL(cn::dbscan_count_kernel, grid, block, 0, s, G, eps2, (int)min_points, neighbour_count, parent);
  hipLaunchKernelGGL(cn::dbscan_union_kernel, grid, block, 0, s, G, eps2, (int)min_points, neighbour_count, parent);
  hipLaunchKernelGGL(cn::dbscan_assign_kernel, grid, block, 0, s, G, eps2, (int)min_points, neighbour_count, parent, order,
                     root);
  return cn::check_launch("cn_dbscan");
}

// ---------------------------------------------------------------------------------------------------------------
// K-means sub-clustering of one super-cluster (segmentation/segmenter.py:28-45,153-181: sklearn KMeans(init="k-means++",
// n_clusters=k, n_init="auto", random_state=0)).  The k-means++ seeding (a handful of weighted random draws) stays on the
// host with scikit-learn's own routine and random stream; this is ONE Lloyd iteration of sklearn's _kmeans_single_lloyd in
// float64 like sklearn: label every point with its nearest centre (first minimum, as np.argmin), accumulate per-cluster
// coordinate sums and counts, and flag whether any label changed (sklearn's strict-convergence test).  The host divides
// the sums and applies the stopping rules (two scalars read back per iteration; a super-cluster converges in ~10-30).
// ---------------------------------------------------------------------------------------------------------------
namespace cn {

constexpr int KM_MAX_K = 32;

__global__ void __launch_bounds__(256)
kmeans_step_kernel(const double* __restrict__ pts, long long n, const double* __restrict__ centers, int k,
                   int* __restrict__ labels, double* __restrict__ sums /*[k,3]*/, long long* __restrict__ counts /*[k]*/,
                   int* __restrict__ changed, int accumulate) {
  __shared__ double c[KM_MAX_K * 3];
  __shared__ double bs[KM_MAX_K * 3];
  __shared__ unsigned long long bc[KM_MAX_K];
  for (int i = threadIdx.x; i < 3 * k; i += blockDim.x) {
    c[i] = centers[i];
    bs[i] = 0.0;
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) bc[i] = 0ull;
  __syncthreads();
  int any = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    int best = 0;
    double bd = 0.0;
    for (int j = 0; j < k; ++j) {
      const double dx = x - c[3 * j], dy = y - c[3 * j + 1], dz = z - c[3 * j + 2];
      const double d = dx * dx + dy * dy + dz * dz;
      if (j == 0 || d < bd) {  // strict <: the first minimum wins
        bd = d;
        best = j;
      }
    }
    any |= labels[i] != best;
    labels[i] = best;
    if (accumulate) {
      atomicAdd(&bs[3 * best], x);
      atomicAdd(&bs[3 * best + 1], y);
      atomicAdd(&bs[3 * best + 2], z);
      atomicAdd(&bc[best], 1ull);
    }
  }
  if (any) atomicOr(changed, 1);
  __syncthreads();
  if (accumulate) {
    for (int i = threadIdx.x; i < 3 * k; i += blockDim.x)
      if (bs[i] != 0.0) atomicAdd(&sums[i], bs[i]);
    for (int i = threadIdx.x; i < k; i += blockDim.x)
      if (bc[i]) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[i]), bc[i]);
  }
}

}  // namespace cn

extern "C" int cn_kmeans_step(const double* points, int64_t num_points, const double* centers, int32_t k, int32_t* labels,
                              double* sums, int64_t* counts, int32_t* changed, int32_t accumulate, cn_stream_t stream) {
  CN_REQUIRE(points && centers && labels && changed, CN_ERR_INVALID, "cn_kmeans_step: null argument");
  CN_REQUIRE(!accumulate || (sums && counts), CN_ERR_INVALID, "cn_kmeans_step: sums / counts required when accumulating");
  CN_REQUIRE(k >= 1 && k <= cn::KM_MAX_K, CN_ERR_UNSUPPORTED, "cn_kmeans_step: k = %d (1..%d)", k, cn::KM_MAX_K);
  if (num_points <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::kmeans_step_kernel, dim3(cn::grid_for(num_points, 256, 1024)), dim3(256), 0, cn::as_stream(stream),
                     points, (long long)num_points, centers, k, labels, sums, reinterpret_cast<long long*>(counts), changed,
                     accumulate);
  return cn::check_launch("cn_kmeans_step");
}

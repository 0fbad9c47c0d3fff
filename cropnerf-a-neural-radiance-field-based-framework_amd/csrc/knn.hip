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

// ---- the two-level grid (cn_point_grid; round 5) -----------------------------------------------------------------------------------
// An exported cloud is not a volume: 10^7 kept points are surfaces, or -- the C4 bench's synthetic cloud -- a few hundred clumps
// with 27 000 points to a cell of any dense grid that fits memory (3.7 s for the outlier pass, 8.1 s for the normals).  Two levels:
// a dense TOP grid (<= 128 cells per axis) whose occupied cells are numbered (top_rank), and sub^3 FINE cells inside every
// occupied top cell, so that fine cells exist only where points are: fine cell (fx, fy, fz) = top cell (f / sub), sub cell
// (f % sub), its points [cell_start[rank * sub^3 + sub index], cell_start[... + 1]).  The points are sorted by that key, so a
// top cell's points are contiguous too -- which is what the search falls back on where the fine rings run out of points: after
// `fine_rings` rings it starts over on the top grid (an isolated point would otherwise walk 2 048^3 empty fine cells).
struct PointGrid {
  int tx, ty, tz, sub, sub3, fine_rings;
  float ox, oy, oz, H, h, inv_h, inv_H;
  const int* __restrict__ top_rank;
  const int* __restrict__ cell_start;
};

// Insert candidate (d, id) into the sorted lists (ascending d); WITH_IDX = false keeps distances only.
template <int K, bool WITH_IDX>
__device__ __forceinline__ void knn_insert(float (&best)[K], int (&bidx)[WITH_IDX ? K : 1], float d, int id) {
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool lt = d < best[j];
    const float keep = lt ? d : best[j];
    d = lt ? best[j] : d;
    best[j] = keep;
    if (WITH_IDX) {
      const int keep_id = lt ? id : bidx[j];
      id = lt ? bidx[j] : id;
      bidx[j] = keep_id;
    }
  }
}

template <int K, bool WITH_IDX>
__device__ __forceinline__ void knn_scan(const float* __restrict__ pts, int lo, int hi, float px, float py, float pz,
                                         float (&best)[K], int (&bidx)[WITH_IDX ? K : 1]) {
  for (int q = lo; q < hi; ++q) {
    const float ex = pts[3 * (long long)q] - px, ey = pts[3 * (long long)q + 1] - py, ez = pts[3 * (long long)q + 2] - pz;
    const float d = ex * ex + ey * ey + ez * ez;
    if (d < best[K - 1]) knn_insert<K, WITH_IDX>(best, bidx, d, q);
  }
}

// the k_used nearest points of (px, py, pz) (squared distances ascending in best[]; indices in bidx[] with WITH_IDX)
template <int K, bool WITH_IDX>
__device__ __forceinline__ void knn_search(const PointGrid& g, const float* __restrict__ pts, float px, float py, float pz, int k_used,
                                           float (&best)[K], int (&bidx)[WITH_IDX ? K : 1]) {
  const int kk = k_used - 1 < K ? k_used - 1 : K - 1;
  auto reset = [&]() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      best[j] = 3.0e38f;
      if (WITH_IDX) bidx[j] = -1;
    }
  };
  reset();
  const int fx_n = g.tx * g.sub, fy_n = g.ty * g.sub, fz_n = g.tz * g.sub;
  const int cx = min(max((int)floorf((px - g.ox) * g.inv_h), 0), fx_n - 1);
  const int cy = min(max((int)floorf((py - g.oy) * g.inv_h), 0), fy_n - 1);
  const int cz = min(max((int)floorf((pz - g.oz) * g.inv_h), 0), fz_n - 1);
  bool done = false;
  for (int ring = 0; ring <= g.fine_rings && !done; ++ring) {
    // the shell of fine cells at Chebyshev distance `ring` from the query's cell
    for (int dz = -ring; dz <= ring; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= fz_n) continue;
      for (int dy = -ring; dy <= ring; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= fy_n) continue;
        const bool face = (dz == -ring || dz == ring || dy == -ring || dy == ring);
        const int step = face ? 1 : max(2 * ring, 1);  // interior rows of the shell: only the two end cells
        const int tzy = ((z / g.sub) * g.ty + (y / g.sub)) * g.tx;
        const int szy = ((z % g.sub) * g.sub + (y % g.sub)) * g.sub;
        for (int dx = -ring; dx <= ring; dx += step) {
          const int x = cx + dx;
          if (x < 0 || x >= fx_n) continue;
          const int r = g.top_rank[tzy + x / g.sub];
          if (r < 0) continue;
          const long long c = (long long)r * g.sub3 + szy + x % g.sub;
          knn_scan<K, WITH_IDX>(pts, g.cell_start[c], g.cell_start[c + 1], px, py, pz, best, bidx);
        }
      }
    }
    // everything outside the (2 ring + 1)^3 block is at least ring * h away (the point lies inside the centre cell)
    const float reach = (float)ring * g.h;
    done = ring >= 1 && best[kk] <= reach * reach;
  }
  if (done) return;
  // not enough points within the fine rings: the same search on the top grid (whole top cells, contiguous point ranges)
  reset();
  const int tcx = cx / g.sub, tcy = cy / g.sub, tcz = cz / g.sub;
  const int max_ring = max(g.tx, max(g.ty, g.tz));
  for (int ring = 0; ring <= max_ring; ++ring) {
    for (int dz = -ring; dz <= ring; ++dz) {
      const int z = tcz + dz;
      if (z < 0 || z >= g.tz) continue;
      for (int dy = -ring; dy <= ring; ++dy) {
        const int y = tcy + dy;
        if (y < 0 || y >= g.ty) continue;
        const bool face = (dz == -ring || dz == ring || dy == -ring || dy == ring);
        const int step = face ? 1 : max(2 * ring, 1);
        for (int dx = -ring; dx <= ring; dx += step) {
          const int x = tcx + dx;
          if (x < 0 || x >= g.tx) continue;
          const int r = g.top_rank[(z * g.ty + y) * g.tx + x];
          if (r < 0) continue;
          knn_scan<K, WITH_IDX>(pts, g.cell_start[(long long)r * g.sub3], g.cell_start[(long long)(r + 1) * g.sub3], px, py, pz, best, bidx);
        }
      }
    }
    const float reach = (float)ring * g.H;
    if (ring >= 1 && best[kk] <= reach * reach) break;
  }
}

template <int K>
__global__ void __launch_bounds__(256)
knn_mean_distance_grid_kernel(PointGrid g, const float* __restrict__ pts, long long n, int k_used, float* __restrict__ mean_out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float best[K];
    int none[1];
    knn_search<K, false>(g, pts, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], k_used, best, none);
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

// ---- normals of the exported cloud ---------------------------------------------------------------------------------------------
// generate_point_cloud estimates normals with open3d's PointCloud::EstimateNormals() at its defaults
// (fruit_nerf/export/exporter_utils_nerfacto.py:203-212; README.md:125 `--normal-method open3d`): per point the `knn` = 30
// nearest points (KDTreeSearchParamKNN(30), the point itself included), their covariance through the nine cumulants
// (sums of x, y, z, xx, xy, xz, yy, yz, zz over the neighbours, divided by their number, in double), and the eigenvector
// of the smallest eigenvalue from the non-iterative solver for symmetric 3 x 3 matrices (Eberly, "A Robust Eigensolver for
// 3 x 3 Symmetric Matrices": open3d's fast_normal_computation = true).  Fewer than three neighbours, or a zero vector from
// the solver: (0, 0, 1), flagged.  Same uniform grid and ring search as the kernel above, keeping the neighbours' indices.
struct Sym3 {
  double a00, a01, a02, a11, a12, a22;
};
__device__ inline void cross3d(const double* u, const double* v, double* w) {
  w[0] = u[1] * v[2] - u[2] * v[1];
  w[1] = u[2] * v[0] - u[0] * v[2];
  w[2] = u[0] * v[1] - u[1] * v[0];
}
// eigenvector of A for a simple eigenvalue e: (A - e I) has rank 2, its rows span the plane orthogonal to the eigenvector --
// the longest of the three row cross products, normalised
__device__ inline void eigvec_of_simple(const Sym3& A, double e, double* out) {
  const double r0[3] = {A.a00 - e, A.a01, A.a02}, r1[3] = {A.a01, A.a11 - e, A.a12}, r2[3] = {A.a02, A.a12, A.a22 - e};
  double c01[3], c02[3], c12[3];
  cross3d(r0, r1, c01);
  cross3d(r0, r2, c02);
  cross3d(r1, r2, c12);
  const double d01 = c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2];
  const double d02 = c02[0] * c02[0] + c02[1] * c02[1] + c02[2] * c02[2];
  const double d12 = c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2];
  double dmax = d01;
  const double* best = c01;
  if (d02 > dmax) {
    dmax = d02;
    best = c02;
  }
  if (d12 > dmax) {
    dmax = d12;
    best = c12;
  }
  const double inv = 1.0 / sqrt(dmax);
  out[0] = best[0] * inv;
  out[1] = best[1] * inv;
  out[2] = best[2] * inv;
}
// eigenvector for the eigenvalue e1 inside the plane orthogonal to the unit vector w (an eigenvector already found): with an
// orthonormal basis (u, v) of that plane, the 2 x 2 matrix [u v]^T A [u v] - e1 I is singular and its kernel gives the
// combination of u and v
__device__ inline void eigvec_in_complement(const Sym3& A, const double* w, double e1, double* out) {
  double u[3], v[3];
  if (fabs(w[0]) > fabs(w[1])) {
    const double inv = 1.0 / sqrt(w[0] * w[0] + w[2] * w[2]);
    u[0] = -w[2] * inv;
    u[1] = 0.0;
    u[2] = w[0] * inv;
  } else {
    const double inv = 1.0 / sqrt(w[1] * w[1] + w[2] * w[2]);
    u[0] = 0.0;
    u[1] = w[2] * inv;
    u[2] = -w[1] * inv;
  }
  cross3d(w, u, v);
  const double au[3] = {A.a00 * u[0] + A.a01 * u[1] + A.a02 * u[2], A.a01 * u[0] + A.a11 * u[1] + A.a12 * u[2],
                        A.a02 * u[0] + A.a12 * u[1] + A.a22 * u[2]};
  const double av[3] = {A.a00 * v[0] + A.a01 * v[1] + A.a02 * v[2], A.a01 * v[0] + A.a11 * v[1] + A.a12 * v[2],
                        A.a02 * v[0] + A.a12 * v[1] + A.a22 * v[2]};
  double m00 = u[0] * au[0] + u[1] * au[1] + u[2] * au[2] - e1;
  double m01 = u[0] * av[0] + u[1] * av[1] + u[2] * av[2];
  double m11 = v[0] * av[0] + v[1] * av[1] + v[2] * av[2] - e1;
  const double am00 = fabs(m00), am01 = fabs(m01), am11 = fabs(m11);
  if (am00 >= am11) {
    if (fmax(am00, am01) > 0.0) {
      if (am00 >= am01) {
        m01 /= m00;
        m00 = 1.0 / sqrt(1.0 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1.0 / sqrt(1.0 + m00 * m00);
        m00 *= m01;
      }
      for (int k = 0; k < 3; ++k) out[k] = m01 * u[k] - m00 * v[k];
    } else {
      for (int k = 0; k < 3; ++k) out[k] = u[k];
    }
  } else {
    if (fmax(am11, am01) > 0.0) {
      if (am11 >= am01) {
        m01 /= m11;
        m11 = 1.0 / sqrt(1.0 + m01 * m01);
        m01 *= m11;
      } else {
        m11 /= m01;
        m01 = 1.0 / sqrt(1.0 + m11 * m11);
        m11 *= m01;
      }
      for (int k = 0; k < 3; ++k) out[k] = m11 * u[k] - m01 * v[k];
    } else {
      for (int k = 0; k < 3; ++k) out[k] = u[k];
    }
  }
}
// unit eigenvector of the smallest eigenvalue of the covariance C (zero vector when C is zero)
__device__ inline void smallest_eigenvector(Sym3 C, double* n) {
  double mx = fmax(fmax(fmax(C.a00, C.a01), fmax(C.a02, C.a11)), fmax(C.a12, C.a22));
  n[0] = n[1] = n[2] = 0.0;
  if (mx == 0.0) return;
  const double s = 1.0 / mx;
  Sym3 A{C.a00 * s, C.a01 * s, C.a02 * s, C.a11 * s, C.a12 * s, C.a22 * s};
  const double off = A.a01 * A.a01 + A.a02 * A.a02 + A.a12 * A.a12;
  if (off > 0.0) {
    const double q = (A.a00 + A.a11 + A.a22) / 3.0;
    const double b00 = A.a00 - q, b11 = A.a11 - q, b22 = A.a22 - q;
    const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * off) / 6.0);
    const double c00 = b11 * b22 - A.a12 * A.a12, c01 = A.a01 * b22 - A.a12 * A.a02, c02 = A.a01 * A.a12 - b11 * A.a02;
    const double det = (b00 * c00 - A.a01 * c01 + A.a02 * c02) / (p * p * p);
    const double half = fmin(fmax(0.5 * det, -1.0), 1.0);
    const double angle = acos(half) / 3.0;
    const double beta2 = 2.0 * cos(angle), beta0 = 2.0 * cos(angle + 2.09439510239319549), beta1 = -(beta0 + beta2);
    const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;  // e0 <= e1 <= e2 up to rounding
    double va[3], vb[3];
    if (half >= 0.0) {  // e2 is the eigenvalue best separated from the others: start there
      eigvec_of_simple(A, e2, va);
      if (e2 < e0 && e2 < e1) {
        n[0] = va[0], n[1] = va[1], n[2] = va[2];
        return;
      }
      eigvec_in_complement(A, va, e1, vb);
      if (e1 < e0 && e1 < e2) {
        n[0] = vb[0], n[1] = vb[1], n[2] = vb[2];
        return;
      }
      cross3d(vb, va, n);
    } else {
      eigvec_of_simple(A, e0, va);
      if (e0 < e1 && e0 < e2) {
        n[0] = va[0], n[1] = va[1], n[2] = va[2];
        return;
      }
      eigvec_in_complement(A, va, e1, vb);
      if (e1 < e0 && e1 < e2) {
        n[0] = vb[0], n[1] = vb[1], n[2] = vb[2];
        return;
      }
      cross3d(va, vb, n);
    }
  } else {  // diagonal already
    if (C.a00 < C.a11 && C.a00 < C.a22) n[0] = 1.0;
    else if (C.a11 < C.a00 && C.a11 < C.a22) n[1] = 1.0;
    else n[2] = 1.0;
  }
}

// covariance of the neighbours listed in bidx (cumulants in double, as open3d) and its smallest eigenvector; returns 1 where
// fewer than three neighbours exist or the solver gives no direction (the normal is (0, 0, 1) then)
template <int K>
__device__ __forceinline__ int normal_of_neighbours(const float* __restrict__ pts, const int (&bidx)[K], int k_used, double* nv) {
  double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (j < k_used && bidx[j] >= 0) {
      const double x = pts[3 * (long long)bidx[j]], y = pts[3 * (long long)bidx[j] + 1], z = pts[3 * (long long)bidx[j] + 2];
      sx += x, sy += y, sz += z;
      sxx += x * x, sxy += x * y, sxz += x * z, syy += y * y, syz += y * z, szz += z * z;
      ++cnt;
    }
  }
  nv[0] = nv[1] = nv[2] = 0.0;
  int flag = 0;
  if (cnt >= 3) {
    const double m = (double)cnt;  // (divisions, as open3d's `cumulants /= n`: coincident points give an exactly zero matrix)
    sx /= m, sy /= m, sz /= m;
    Sym3 C{sxx / m - sx * sx, sxy / m - sx * sy, sxz / m - sx * sz, syy / m - sy * sy, syz / m - sy * sz,
           szz / m - sz * sz};
    smallest_eigenvector(C, nv);
  }
  if (nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2] == 0.0) {
    nv[2] = 1.0;
    flag = 1;
  }
  return flag;
}

template <int K>
__global__ void __launch_bounds__(256)
knn_normals_kernel(const float* __restrict__ pts /*sorted by cell*/, const int* __restrict__ cell_start, int gx, int gy, int gz,
                   float ox, float oy, float oz, float inv_h, float h, long long n, int k_used, int max_ring,
                   double* __restrict__ normals, int* __restrict__ flags) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int cx = min(max((int)floorf((px - ox) * inv_h), 0), gx - 1);
    const int cy = min(max((int)floorf((py - oy) * inv_h), 0), gy - 1);
    const int cz = min(max((int)floorf((pz - oz) * inv_h), 0), gz - 1);
    float best[K];
    int bidx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      best[j] = 3.0e38f;
      bidx[j] = -1;
    }
    for (int ring = 0; ring <= max_ring; ++ring) {
      for (int dz = -ring; dz <= ring; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= gz) continue;
        for (int dy = -ring; dy <= ring; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y >= gy) continue;
          const bool face = (dz == -ring || dz == ring || dy == -ring || dy == ring);
          const int step = face ? 1 : max(2 * ring, 1);
          for (int dx = -ring; dx <= ring; dx += step) {
            const int x = cx + dx;
            if (x < 0 || x >= gx) continue;
            const long long c = ((long long)z * gy + y) * gx + x;
            const int lo = cell_start[c], hi = cell_start[c + 1];
            for (int q = lo; q < hi; ++q) {
              const float ex = pts[3 * q] - px, ey = pts[3 * q + 1] - py, ez = pts[3 * q + 2] - pz;
              float d = ex * ex + ey * ey + ez * ez;
              if (d < best[K - 1]) {
                int id = q;
#pragma unroll
                for (int j = 0; j < K; ++j) {  // sorted insertion of (d, id), fully unrolled
                  const bool lt = d < best[j];
                  const float keep = lt ? d : best[j];
                  const int keep_id = lt ? id : bidx[j];
                  d = lt ? best[j] : d;
                  id = lt ? bidx[j] : id;
                  best[j] = keep;
                  bidx[j] = keep_id;
                }
              }
            }
          }
        }
      }
      const float reach = (float)ring * h;
      if (best[k_used - 1 < K ? k_used - 1 : K - 1] <= reach * reach && ring >= 1) break;
    }
    double nv[3];
    const int flag = normal_of_neighbours<K>(pts, bidx, k_used, nv);
    normals[3 * i] = nv[0];
    normals[3 * i + 1] = nv[1];
    normals[3 * i + 2] = nv[2];
    if (flags) flags[i] = flag;
  }
}

template <int K>
__global__ void __launch_bounds__(256)
knn_normals_grid_kernel(PointGrid g, const float* __restrict__ pts, long long n, int k_used, double* __restrict__ normals,
                        int* __restrict__ flags) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float best[K];
    int bidx[K];
    knn_search<K, true>(g, pts, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], k_used, best, bidx);
    double nv[3];
    const int flag = normal_of_neighbours<K>(pts, bidx, k_used, nv);
    normals[3 * i] = nv[0];
    normals[3 * i + 1] = nv[1];
    normals[3 * i + 2] = nv[2];
    if (flags) flags[i] = flag;
  }
}

static int make_point_grid(const cn_point_grid* grid, PointGrid* out, const char* who) {
  CN_REQUIRE(grid && grid->top_rank && grid->cell_start, CN_ERR_INVALID, "%s: null grid", who);
  CN_REQUIRE(grid->top[0] > 0 && grid->top[1] > 0 && grid->top[2] > 0 && grid->top[0] <= 1024 && grid->top[1] <= 1024 &&
                 grid->top[2] <= 1024 && grid->sub >= 1 && grid->sub <= 64 && grid->top_cell_size > 0.f && grid->fine_rings >= 1,
             CN_ERR_INVALID, "%s: bad grid (top %d x %d x %d, sub %d, top cell %g, fine rings %d)", who, grid->top[0], grid->top[1],
             grid->top[2], grid->sub, (double)grid->top_cell_size, grid->fine_rings);
  PointGrid g;
  g.tx = grid->top[0];
  g.ty = grid->top[1];
  g.tz = grid->top[2];
  g.sub = grid->sub;
  g.sub3 = grid->sub * grid->sub * grid->sub;
  g.fine_rings = grid->fine_rings;
  g.ox = grid->origin[0];
  g.oy = grid->origin[1];
  g.oz = grid->origin[2];
  g.H = grid->top_cell_size;
  g.h = grid->top_cell_size / (float)grid->sub;
  g.inv_h = 1.f / g.h;
  g.inv_H = 1.f / g.H;
  g.top_rank = grid->top_rank;
  g.cell_start = grid->cell_start;
  *out = g;
  return CN_OK;
}

}  // namespace cn

extern "C" int cn_knn_mean_distance_grid(const float* points_sorted, const cn_point_grid* grid, int64_t num_points,
                                         int32_t nb_neighbors, float* mean_distance, cn_stream_t stream) {
  CN_REQUIRE(nb_neighbors >= 1 && nb_neighbors <= cn::KNN_MAX_K, CN_ERR_UNSUPPORTED,
             "cn_knn_mean_distance_grid: nb_neighbors %d (max %d)", nb_neighbors, cn::KNN_MAX_K);
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(points_sorted && mean_distance, CN_ERR_INVALID, "cn_knn_mean_distance_grid: null argument");
  cn::PointGrid g;
  if (int rc = cn::make_point_grid(grid, &g, "cn_knn_mean_distance_grid")) return rc;
  hipLaunchKernelGGL(cn::knn_mean_distance_grid_kernel<cn::KNN_MAX_K>, dim3(cn::grid_for(num_points, 256, 1 << 16)), dim3(256), 0,
                     cn::as_stream(stream), g, points_sorted, (long long)num_points, nb_neighbors, mean_distance);
  return cn::check_launch("cn_knn_mean_distance_grid");
}

extern "C" int cn_estimate_normals_grid(const float* points_sorted, const cn_point_grid* grid, int64_t num_points, int32_t knn,
                                        double* normals, int32_t* degenerate, cn_stream_t stream) {
  CN_REQUIRE(knn >= 1 && knn <= cn::KNN_MAX_K, CN_ERR_UNSUPPORTED, "cn_estimate_normals_grid: knn %d (max %d)", knn, cn::KNN_MAX_K);
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(points_sorted && normals, CN_ERR_INVALID, "cn_estimate_normals_grid: null argument");
  cn::PointGrid g;
  if (int rc = cn::make_point_grid(grid, &g, "cn_estimate_normals_grid")) return rc;
  hipLaunchKernelGGL(cn::knn_normals_grid_kernel<cn::KNN_MAX_K>, dim3(cn::grid_for(num_points, 256, 1 << 16)), dim3(256), 0,
                     cn::as_stream(stream), g, points_sorted, (long long)num_points, knn, normals, degenerate);
  return cn::check_launch("cn_estimate_normals_grid");
}

extern "C" int cn_estimate_normals(const float* points_sorted, const int32_t* cell_start, int32_t gx, int32_t gy, int32_t gz,
                                   float origin_x, float origin_y, float origin_z, float cell_size, int64_t num_points,
                                   int32_t knn, double* normals, int32_t* degenerate, cn_stream_t stream) {
  CN_REQUIRE(gx > 0 && gy > 0 && gz > 0 && cell_size > 0.f, CN_ERR_INVALID, "cn_estimate_normals: bad grid");
  CN_REQUIRE(knn >= 1 && knn <= cn::KNN_MAX_K, CN_ERR_UNSUPPORTED, "cn_estimate_normals: knn %d (max %d)", knn, cn::KNN_MAX_K);
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(points_sorted && cell_start && normals, CN_ERR_INVALID, "cn_estimate_normals: null argument");
  const int max_ring = std::max(gx, std::max(gy, gz));
  hipLaunchKernelGGL(cn::knn_normals_kernel<cn::KNN_MAX_K>, dim3(cn::grid_for(num_points, 256, 1 << 16)), dim3(256), 0,
                     cn::as_stream(stream), points_sorted, cell_start, gx, gy, gz, origin_x, origin_y, origin_z, 1.f / cell_size,
                     cell_size, (long long)num_points, knn, max_ring, normals, degenerate);
  return cn::check_launch("cn_estimate_normals");
}

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

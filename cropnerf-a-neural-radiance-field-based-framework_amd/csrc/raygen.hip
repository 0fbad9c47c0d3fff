// Ray generation kernels: pinhole cameras, ray/AABB slab test, orthographic surface rays, SO3xR3 pose
// refinement.  One thread per ray, SoA outputs (coalesced 4-byte stores per component row).
//
// Reference call sites (crop_nerf/...): fruit_nerf/data/fruit_datamanager.py:188-197, fruit_nerf/fruit_nerf.py:283-286,
// fruit_nerf/export/exporter_utils_nerfacto.py:266-268, fruit_nerf/components/ray_generators.py:46-66,
// fruit_nerf/data/fruit_datamanager.py:71-121, fruit_nerf/fruit_nerf.py:547.
#include "cn_common.hpp"
#include "raygen_dev.hpp"

namespace cn {

__global__ void __launch_bounds__(256)
raygen_pinhole_kernel(const float* __restrict__ c2w, const float* __restrict__ intr,
                      const int64_t* __restrict__ ray_indices, int cam, int height, int width, long long pixel_start,
                      long long num_rays, int cam_value, float* __restrict__ origins, float* __restrict__ directions,
                      float* __restrict__ pixel_area, int64_t* __restrict__ camera_indices,
                      float* __restrict__ directions_norm) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < num_rays;
       r += (long long)gridDim.x * blockDim.x) {
    long long c, row, col;
    if (ray_indices) {
      c = ray_indices[3 * r + 0];
      row = ray_indices[3 * r + 1];
      col = ray_indices[3 * r + 2];
    } else {
      long long pix = pixel_start + r;
      c = cam;
      row = pix / width;
      col = pix % width;
    }
    const float* m = c2w + 12 * c;
    float fx = intr[4 * c + 0], fy = intr[4 * c + 1], px = intr[4 * c + 2], py = intr[4 * c + 3];
    float y = (float)row + 0.5f, x = (float)col + 0.5f;
    float cx0, cy0;
    pixel_camera_coords(fx, fy, px, py, row, col, cx0, cy0);
    float cx1 = (x - px + 1.f) / fx, cy1 = -(y - py + 1.f) / fy;
    float d0x, d0y, d0z, n0, d1x, d1y, d1z, n1, d2x, d2y, d2z, n2;
    rotate_normalize(m, cx0, cy0, d0x, d0y, d0z, n0);
    rotate_normalize(m, cx1, cy0, d1x, d1y, d1z, n1);
    rotate_normalize(m, cx0, cy1, d2x, d2y, d2z, n2);
    if (origins) {
      origins[3 * r + 0] = m[3];
      origins[3 * r + 1] = m[7];
      origins[3 * r + 2] = m[11];
    }
    if (directions) {
      directions[3 * r + 0] = d0x;
      directions[3 * r + 1] = d0y;
      directions[3 * r + 2] = d0z;
    }
    if (pixel_area) {
      float ax = d0x - d1x, ay = d0y - d1y, az = d0z - d1z;
      float bx = d0x - d2x, by = d0y - d2y, bz = d0z - d2z;
      pixel_area[r] = sqrtf(ax * ax + ay * ay + az * az) * sqrtf(bx * bx + by * by + bz * bz);
    }
    if (camera_indices) camera_indices[r] = cam_value >= 0 ? (int64_t)cam_value : (int64_t)c;
    if (directions_norm) directions_norm[r] = n0;
  }
}

__global__ void __launch_bounds__(256)
intersect_aabb_kernel(const float* __restrict__ o, const float* __restrict__ d, float lx, float ly, float lz, float hx,
                      float hy, float hz, long long n, float* __restrict__ nears, float* __restrict__ fars) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
    float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2];
    float dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
    float tn, tf;
    slab_test(ox, oy, oz, dx, dy, dz, lx, ly, lz, hx, hy, hz, tn, tf);
    nears[r] = tn;
    fars[r] = tf;
  }
}

__global__ void __launch_bounds__(256)
raygen_ortho_kernel(const float* __restrict__ pts, float nx, float ny, float nz, float len, long long start,
                    long long n, float* __restrict__ origins, float* __restrict__ directions,
                    float* __restrict__ pixel_area, float* __restrict__ nears, float* __restrict__ fars) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
    const float* p = pts + 3 * (start + r);
    origins[3 * r + 0] = p[0];
    origins[3 * r + 1] = p[1];
    origins[3 * r + 2] = p[2];
    directions[3 * r + 0] = nx;
    directions[3 * r + 1] = ny;
    directions[3 * r + 2] = nz;
    if (pixel_area) pixel_area[r] = 0.f;
    nears[r] = 0.f;
    fars[r] = len;
  }
}

// torch.linspace(a, b, n)[i] as the CPU kernel computes it
__device__ __forceinline__ float linspace_ab(float a, float b, int i, int n) {
  if (n == 1) return a;
  float step = (b - a) / (float)(n - 1);
  return i < n / 2 ? a + step * (float)i : b - step * (float)(n - i - 1);
}

__global__ void __launch_bounds__(256)
surface_grid_kernel(float x0, float x1, int nx, float y0, float y1, int ny, float z, float* __restrict__ pts) {
  long long total = (long long)nx * ny;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int ix = (int)(i / ny), iy = (int)(i % ny);
    pts[3 * i + 0] = linspace_ab(x0, x1, ix, nx);
    pts[3 * i + 1] = linspace_ab(y0, y1, iy, ny);
    pts[3 * i + 2] = z;
  }
}

__global__ void __launch_bounds__(256)
pose_adjust_kernel(const float* __restrict__ adj, const int64_t* __restrict__ cam, long long n, const float* o_in,
                   const float* d_in, float* o, float* d) {  // o_in / d_in may alias o / d (the in-place form)
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
    const float* a = adj + 6 * cam[r];
    float tx = a[0], ty = a[1], tz = a[2], wx = a[3], wy = a[4], wz = a[5];
    float nrm = wx * wx + wy * wy + wz * wz;
    float ang = sqrtf(fmaxf(nrm, 1e-4f));
    float inv = 1.f / ang;
    float f1 = inv * sinf(ang);
    float f2 = inv * inv * (1.f - cosf(ang));
    // skew K and K^2
    float k01 = -wz, k02 = wy, k10 = wz, k12 = -wx, k20 = -wy, k21 = wx;
    float s00 = k01 * k10 + k02 * k20, s01 = k02 * k21, s02 = k01 * k12;
    float s10 = k12 * k20, s11 = k10 * k01 + k12 * k21, s12 = k10 * k02;
    float s20 = k21 * k10, s21 = k20 * k01, s22 = k20 * k02 + k21 * k12;
    float r00 = f2 * s00 + 1.f, r01 = f1 * k01 + f2 * s01, r02 = f1 * k02 + f2 * s02;
    float r10 = f1 * k10 + f2 * s10, r11 = f2 * s11 + 1.f, r12 = f1 * k12 + f2 * s12;
    float r20 = f1 * k20 + f2 * s20, r21 = f1 * k21 + f2 * s21, r22 = f2 * s22 + 1.f;
    float dx = d_in[3 * r], dy = d_in[3 * r + 1], dz = d_in[3 * r + 2];
    o[3 * r + 0] = o_in[3 * r + 0] + tx;
    o[3 * r + 1] = o_in[3 * r + 1] + ty;
    o[3 * r + 2] = o_in[3 * r + 2] + tz;
    d[3 * r + 0] = r00 * dx + r01 * dy + r02 * dz;
    d[3 * r + 1] = r10 * dx + r11 * dy + r12 * dz;
    d[3 * r + 2] = r20 * dx + r21 * dy + r22 * dz;
  }
}

__global__ void __launch_bounds__(64)
embedding_mean_kernel(const float* __restrict__ emb, int n, int dim, float* __restrict__ mean) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dim) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += emb[(long long)i * dim + j];
  mean[j] = s / (float)n;
}

}  // namespace cn

extern "C" int cn_raygen_pinhole(const float* c2w, const float* intrinsics, const int64_t* ray_indices, int32_t cam,
                                 int32_t height, int32_t width, int64_t pixel_start, int64_t num_rays,
                                 int32_t camera_index_value, float* origins, float* directions, float* pixel_area,
                                 int64_t* camera_indices, float* directions_norm, cn_stream_t stream) {
  CN_REQUIRE(c2w && intrinsics, CN_ERR_INVALID, "cn_raygen_pinhole: null camera arrays");
  CN_REQUIRE(num_rays >= 0, CN_ERR_INVALID, "cn_raygen_pinhole: negative num_rays");
  if (!ray_indices) {
    CN_REQUIRE(height > 0 && width > 0 && cam >= 0, CN_ERR_INVALID, "cn_raygen_pinhole: bad image size / camera");
    CN_REQUIRE(pixel_start >= 0 && pixel_start + num_rays <= (int64_t)height * width, CN_ERR_INVALID,
               "cn_raygen_pinhole: pixel range [%lld,%lld) outside %dx%d", (long long)pixel_start,
               (long long)(pixel_start + num_rays), height, width);
  }
  if (num_rays == 0) return CN_OK;
  hipLaunchKernelGGL(cn::raygen_pinhole_kernel, dim3(cn::grid_for(num_rays, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), c2w, intrinsics, ray_indices, cam, height, width, (long long)pixel_start,
                     (long long)num_rays, camera_index_value, origins, directions, pixel_area, camera_indices,
                     directions_norm);
  return cn::check_launch("cn_raygen_pinhole");
}

extern "C" int cn_intersect_aabb(const float* origins, const float* directions, const float* aabb, int64_t num_rays,
                                 float* nears, float* fars, cn_stream_t stream) {
  CN_REQUIRE(origins && directions && aabb && nears && fars, CN_ERR_INVALID, "cn_intersect_aabb: null argument");
  if (num_rays <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::intersect_aabb_kernel, dim3(cn::grid_for(num_rays, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), origins, directions, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5],
                     (long long)num_rays, nears, fars);
  return cn::check_launch("cn_intersect_aabb");
}

extern "C" int cn_raygen_ortho(const float* surface_points, const float* plane_vector, int64_t start,
                               int64_t num_rays, float* origins, float* directions, float* pixel_area, float* nears,
                               float* fars, cn_stream_t stream) {
  CN_REQUIRE(surface_points && plane_vector && origins && directions && nears && fars, CN_ERR_INVALID,
             "cn_raygen_ortho: null argument");
  CN_REQUIRE(start >= 0, CN_ERR_INVALID, "cn_raygen_ortho: negative start");
  if (num_rays <= 0) return CN_OK;
  float x = plane_vector[0], y = plane_vector[1], z = plane_vector[2];
  float len = sqrtf(x * x + y * y + z * z);
  float den = fmaxf(len, 1e-12f);  // torch.nn.functional.normalize eps
  hipLaunchKernelGGL(cn::raygen_ortho_kernel, dim3(cn::grid_for(num_rays, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), surface_points, x / den, y / den, z / den, len, (long long)start,
                     (long long)num_rays, origins, directions, pixel_area, nears, fars);
  return cn::check_launch("cn_raygen_ortho");
}

extern "C" int cn_surface_grid(float x0, float x1, int32_t nx, float y0, float y1, int32_t ny, float z_const,
                               float* surface_points, cn_stream_t stream) {
  CN_REQUIRE(surface_points && nx > 0 && ny > 0, CN_ERR_INVALID, "cn_surface_grid: bad argument");
  hipLaunchKernelGGL(cn::surface_grid_kernel, dim3(cn::grid_for((long long)nx * ny, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), x0, x1, nx, y0, y1, ny, z_const, surface_points);
  return cn::check_launch("cn_surface_grid");
}

extern "C" int cn_apply_pose_adjustment(const float* pose_adjustment, const int64_t* camera_indices, int64_t num_rays,
                                        float* origins, float* directions, cn_stream_t stream) {
  CN_REQUIRE(pose_adjustment && camera_indices && origins && directions, CN_ERR_INVALID,
             "cn_apply_pose_adjustment: null argument");
  if (num_rays <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::pose_adjust_kernel, dim3(cn::grid_for(num_rays, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), pose_adjustment, camera_indices, (long long)num_rays, origins, directions, origins,
                     directions);
  return cn::check_launch("cn_apply_pose_adjustment");
}

extern "C" int cn_apply_pose_adjustment_to(const float* pose_adjustment, const int64_t* camera_indices, int64_t num_rays,
                                           const float* origins, const float* directions, float* out_origins,
                                           float* out_directions, cn_stream_t stream) {
  CN_REQUIRE(pose_adjustment && camera_indices && origins && directions && out_origins && out_directions, CN_ERR_INVALID,
             "cn_apply_pose_adjustment_to: null argument");
  if (num_rays <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::pose_adjust_kernel, dim3(cn::grid_for(num_rays, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), pose_adjustment, camera_indices, (long long)num_rays, origins, directions,
                     out_origins, out_directions);
  return cn::check_launch("cn_apply_pose_adjustment_to");
}

extern "C" int cn_embedding_mean(const float* embedding, int32_t num_images, int32_t dim, float* mean,
                                 cn_stream_t stream) {
  CN_REQUIRE(embedding && mean && num_images > 0 && dim > 0, CN_ERR_INVALID, "cn_embedding_mean: bad argument");
  hipLaunchKernelGGL(cn::embedding_mean_kernel, dim3((dim + 63) / 64), dim3(64), 0, cn::as_stream(stream), embedding,
                     num_images, dim, mean);
  return cn::check_launch("cn_embedding_mean");
}

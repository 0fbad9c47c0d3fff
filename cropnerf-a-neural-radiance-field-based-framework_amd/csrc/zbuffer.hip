// Depth-based semantic projection: the z-buffer splat of
// fruit_nerf/scripts/depth_based_semantic_projection.py -- get_projection (:45-49) and update_buffer (:84-105) -- as
// atomic splats.  The reference works in float64 (numpy) with a float32 z-buffer; so does this.
//
//   cn_depth_project          im = P @ [p, 1];  yx = round(im[:2] / -im[2]) (half to even, as np.round);
//                             ys = clip(yx[0], 0, W-1), xs = clip(yx[1], 0, H-1), zs = -im[2]
//   cn_zbuffer_update_large   update_buffer(large=True): img[xs, ys] = label; z[xs, ys] = zs -- numpy fancy assignment, the
//                             LAST point of a pixel wins: atomicMax of the point index per pixel, then one write per pixel
//   cn_zbuffer_update         update_buffer(large=False): the sequential "if z <= z_buffer[x, y]" loop.  Its result does not
//                             depend on the point order: a pixel is touched iff min_i z_i <= z_buffer, and then holds that
//                             minimum and the label -- an atomicMin splat on order-preserving integer keys, then one pass
//                             over the pixels.  (Tie caveat, documented in DESIGN.md: the reference compares float64 z with
//                             the float32 buffer it has just rounded into, which can differ by one float32 ulp from this.)
#include "cn_common.hpp"

namespace cn {

__global__ void __launch_bounds__(256)
depth_project_kernel(const double* __restrict__ P, const double* __restrict__ pts, long long n, int H, int W,
                     int* __restrict__ xs, int* __restrict__ ys, double* __restrict__ zs) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    // row-times-column in numpy's order of accumulation (P @ points_h.T, one dot product per output)
    const double a = ((P[0] * x + P[1] * y) + P[2] * z) + P[3];
    const double b = ((P[4] * x + P[5] * y) + P[6] * z) + P[7];
    const double c = ((P[8] * x + P[9] * y) + P[10] * z) + P[11];
    const double u = rint(a / -c), v = rint(b / -c);
    const double uc = fmin(fmax(u, 0.0), (double)(W - 1)), vc = fmin(fmax(v, 0.0), (double)(H - 1));
    ys[i] = (int)uc;  // column
    xs[i] = (int)vc;  // row
    zs[i] = -c;
  }
}

__device__ __forceinline__ unsigned long long ordered_key(double z) {
  unsigned long long b = (unsigned long long)__double_as_longlong(z);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // monotone: smaller double <-> smaller key
}
__device__ __forceinline__ double key_to_double(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ void __launch_bounds__(256) fill_i64_kernel(long long* p, long long n, long long v) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void __launch_bounds__(256)
splat_last_kernel(const int* __restrict__ xs, const int* __restrict__ ys, long long n, int W, long long* __restrict__ last) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    atomicMax(reinterpret_cast<unsigned long long*>(last + (long long)xs[i] * W + ys[i]), (unsigned long long)(i + 1));
}

__global__ void __launch_bounds__(256)
apply_last_kernel(const long long* __restrict__ last, const double* __restrict__ zs, long long npix, int label,
                  float* __restrict__ zbuf, unsigned char* __restrict__ img) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long k = last[p];
    if (k > 0) {
      zbuf[p] = (float)zs[k - 1];
      img[p] = (unsigned char)label;
    }
  }
}

__global__ void __launch_bounds__(256)
splat_min_kernel(const int* __restrict__ xs, const int* __restrict__ ys, const double* __restrict__ zs, long long n, int W,
                 unsigned long long* __restrict__ zmin) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    atomicMin(zmin + (long long)xs[i] * W + ys[i], ordered_key(zs[i]));
}

__global__ void __launch_bounds__(256)
apply_min_kernel(const unsigned long long* __restrict__ zmin, long long npix, int label, float* __restrict__ zbuf,
                 unsigned char* __restrict__ img, unsigned char* __restrict__ visible) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const unsigned long long k = zmin[p];
    unsigned char vis = 0;
    if (k != 0xffffffffffffffffull) {
      const double z = key_to_double(k);
      if (z <= (double)zbuf[p]) {
        zbuf[p] = (float)z;
        img[p] = (unsigned char)label;
        vis = 255;
      }
    }
    if (visible) visible[p] = vis;
  }
}

}  // namespace cn

extern "C" int cn_depth_project(const double* P, const double* points, int64_t num_points, int32_t height,
                                int32_t width, int32_t* xs, int32_t* ys, double* zs, cn_stream_t stream) {
  CN_REQUIRE(height > 0 && width > 0, CN_ERR_INVALID, "cn_depth_project: bad image size");
  if (num_points <= 0) return CN_OK;  // an empty cloud is a no-op (empty tensors carry null data pointers)
  CN_REQUIRE(P && points && xs && ys && zs, CN_ERR_INVALID, "cn_depth_project: null argument");
  hipLaunchKernelGGL(cn::depth_project_kernel, dim3(cn::grid_for(num_points, 256, 8192)), dim3(256), 0,
                     cn::as_stream(stream), P, points, (long long)num_points, height, width, xs, ys, zs);
  return cn::check_launch("cn_depth_project");
}

extern "C" size_t cn_zbuffer_workspace_bytes(int32_t height, int32_t width) {
  return height > 0 && width > 0 ? (size_t)height * width * sizeof(long long) : 0;
}

extern "C" int cn_zbuffer_update_large(const int32_t* xs, const int32_t* ys, const double* zs, int64_t num_points,
                                       int32_t label, int32_t height, int32_t width, float* z_buffer, uint8_t* img,
                                       void* workspace, size_t workspace_bytes, cn_stream_t stream) {
  CN_REQUIRE(z_buffer && img && height > 0 && width > 0, CN_ERR_INVALID, "cn_zbuffer_update_large: bad argument");
  if (num_points <= 0) return CN_OK;
  CN_REQUIRE(xs && ys && zs, CN_ERR_INVALID, "cn_zbuffer_update_large: null input");
  CN_REQUIRE(workspace && workspace_bytes >= cn_zbuffer_workspace_bytes(height, width), CN_ERR_WORKSPACE,
             "cn_zbuffer_update_large: workspace too small");
  const long long npix = (long long)height * width;
  hipStream_t s = cn::as_stream(stream);
  long long* last = static_cast<long long*>(workspace);
  hipLaunchKernelGGL(cn::fill_i64_kernel, dim3(cn::grid_for(npix, 256, 8192)), dim3(256), 0, s, last, npix, 0LL);
  hipLaunchKernelGGL(cn::splat_last_kernel, dim3(cn::grid_for(num_points, 256, 8192)), dim3(256), 0, s, xs, ys,
                     (long long)num_points, width, last);
  hipLaunchKernelGGL(cn::apply_last_kernel, dim3(cn::grid_for(npix, 256, 8192)), dim3(256), 0, s, last, zs, npix,
                     (int)label, z_buffer, img);
  return cn::check_launch("cn_zbuffer_update_large");
}

extern "C" int cn_zbuffer_update(const int32_t* xs, const int32_t* ys, const double* zs, int64_t num_points,
                                 int32_t label, int32_t height, int32_t width, float* z_buffer, uint8_t* img,
                                 uint8_t* visible, void* workspace, size_t workspace_bytes, cn_stream_t stream) {
  CN_REQUIRE(z_buffer && img && height > 0 && width > 0, CN_ERR_INVALID, "cn_zbuffer_update: bad argument");
  CN_REQUIRE(num_points <= 0 || (xs && ys && zs), CN_ERR_INVALID, "cn_zbuffer_update: null input");
  CN_REQUIRE(workspace && workspace_bytes >= cn_zbuffer_workspace_bytes(height, width), CN_ERR_WORKSPACE,
             "cn_zbuffer_update: workspace too small");
  const long long npix = (long long)height * width;
  hipStream_t s = cn::as_stream(stream);
  unsigned long long* zmin = static_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(cn::fill_i64_kernel, dim3(cn::grid_for(npix, 256, 8192)), dim3(256), 0, s,
                     reinterpret_cast<long long*>(zmin), npix, -1LL);
  if (num_points > 0)
    hipLaunchKernelGGL(cn::splat_min_kernel, dim3(cn::grid_for(num_points, 256, 8192)), dim3(256), 0, s, xs, ys, zs,
                       (long long)num_points, width, zmin);
  hipLaunchKernelGGL(cn::apply_min_kernel, dim3(cn::grid_for(npix, 256, 8192)), dim3(256), 0, s, zmin, npix, (int)label,
                     z_buffer, img, visible);
  return cn::check_launch("cn_zbuffer_update");
}

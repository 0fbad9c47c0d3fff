// Per-ray arithmetic of the pinhole camera model and the ray / AABB slab test, shared by the ray-generation kernels
// (raygen.hip) and the batched projection kernels (projection.hip) so that both produce the SAME bits for a pixel.
//
// Reference call sites (crop_nerf/...): fruit_nerf/fruit_nerf.py:283-286 (generate_rays(aabb_box=...)); arithmetic =
// nerfstudio Cameras._generate_rays_from_coords / utils.math.intersect_aabb.
#pragma once

#include "cn_common.hpp"

namespace cn {

__device__ __forceinline__ void rotate_normalize(const float* __restrict__ m /*3x4 row-major*/, float cx, float cy,
                                                 float& dx, float& dy, float& dz, float& norm) {
  // d = sum_j dir[j] * R[i][j], dir = (cx, cy, -1)
  float x = cx * m[0] + cy * m[1] - m[2];
  float y = cx * m[4] + cy * m[5] - m[6];
  float z = cx * m[8] + cy * m[9] - m[10];
  float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-7f);
  dx = x / n;
  dy = y / n;
  dz = z / n;
  norm = n;
}

// camera-frame coordinates of the centre of pixel (row, col): ((x - cx) / fx, -(y - cy) / fy), x = col + 0.5, y = row + 0.5
__device__ __forceinline__ void pixel_camera_coords(float fx, float fy, float px, float py, long long row, long long col,
                                                    float& cx0, float& cy0) {
  float y = (float)row + 0.5f, x = (float)col + 0.5f;
  cx0 = (x - px) / fx;
  cy0 = -(y - py) / fy;
}

// intersect_aabb(o, d, aabb, max_bound = 1e10, invalid = 1e10): misses get 1e10 in both outputs
__device__ __forceinline__ void slab_test(float ox, float oy, float oz, float dx, float dy, float dz, float lx, float ly,
                                          float lz, float hx, float hy, float hz, float& near, float& far) {
  float ax = (lx - ox) / dx, bx = (hx - ox) / dx;
  float ay = (ly - oy) / dy, by = (hy - oy) / dy;
  float az = (lz - oz) / dz, bz = (hz - oz) / dz;
  float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  tmin = fminf(fmaxf(tmin, 0.f), 1e10f);
  tmax = fminf(fmaxf(tmax, 0.f), 1e10f);
  bool miss = tmax <= tmin;
  near = miss ? 1e10f : tmin;
  far = miss ? 1e10f : tmax;
}

}  // namespace cn

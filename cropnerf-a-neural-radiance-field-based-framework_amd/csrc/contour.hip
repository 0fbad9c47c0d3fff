// contour.hip -- the image stage of the merger on the device (segmentation/merger.py:219-271):
// per projected sub-cluster image, threshold -> the contour of largest area -> its area / bounding box
// (get_wo_occlusion_projection_area) and, inside that box of the visible image, the vertex pixels of the largest contour and
// the majority instance label under them (get_visible_projection_area).  The reference runs OpenCV on PNG files, one image
// at a time on the host; here a STACK of J images that are already in HBM (the a15 projection outputs) goes through four
// launches.
//
// OpenCV semantics reproduced (restated in oracle/contours.py, which runs the full Suzuki-Abe raster scan):
//   * foreground = gray > thresh, 8-connected; a contour = the border-following of cv::findContours
//     (icvFetchContour: first neighbour clockwise from west, then counter-clockwise search from the previous pixel), points
//     kept where the step direction changes (CHAIN_APPROX_SIMPLE);
//   * contourArea = |Green's formula| over the border polygon (through pixel centres), boundingRect = extent of its points;
//   * max(contours, key=contourArea): the contour of largest area is always an OUTER border (a hole border lies inside the
//     outer border of its component), so holes are never traced; among equal areas Python's max takes the first contour of
//     OpenCV's list, which holds later-found contours first -> the component whose first pixel comes LAST in raster order;
//   * drawContours(mask, cnt, -1, 255, -1) with a bare contour draws one single-point contour per vertex: the "area" of the
//     visible projection is the number of distinct vertex pixels, and labels are collected under those pixels only.
//
// Decomposition (not OpenCV's): components by lock-free union-find over the pixels (smaller index wins, so a component's
// root IS its first pixel in raster order = the pixel where OpenCV starts that outer border); every root traces its
// border and competes for the job's 64-bit key (area bits | start pixel); one thread per job re-traces the winner.
#include "cn_common.hpp"

namespace cn {

__device__ __forceinline__ int cdx(int s) { return s == 0 || s == 1 || s == 7 ? 1 : (s == 2 || s == 6 ? 0 : -1); }
__device__ __forceinline__ int cdy(int s) { return s >= 1 && s <= 3 ? -1 : (s >= 5 ? 1 : 0); }

struct ContourImg {  // one job's binary predicate: inside the region of interest and above the threshold
  const uint8_t* gray;
  int H, W, x0, y0, x1, y1, thresh;
  __device__ __forceinline__ bool fg(int y, int x) const {
    return x >= x0 && x < x1 && y >= y0 && y < y1 && (int)gray[(long long)y * W + x] > thresh;
  }
};

__device__ __forceinline__ ContourImg contour_img(const uint8_t* gray, const int* roi, int job, int H, int W, int thresh) {
  ContourImg im;
  im.gray = gray + (long long)job * H * W;
  im.H = H;
  im.W = W;
  im.thresh = thresh;
  im.x0 = roi ? max(roi[4 * job], 0) : 0;
  im.y0 = roi ? max(roi[4 * job + 1], 0) : 0;
  im.x1 = roi ? min(roi[4 * job + 2], W) : W;
  im.y1 = roi ? min(roi[4 * job + 3], H) : H;
  return im;
}

// icvFetchContour for an outer border that starts at (y0, x0); VISIT(x, y) is called at every vertex (direction change).
// Returns twice the signed area (Green's formula over the steps) or, when the step budget is exhausted (it cannot be for a
// border that starts at its component's first raster pixel), stops early.
template <typename VISIT>
__device__ __forceinline__ long long trace_outer(const ContourImg& im, int y0, int x0, long long max_steps, VISIT&& visit) {
  int s = 4;
  bool found = false;
  do {
    s = (s - 1) & 7;
    if (im.fg(y0 + cdy(s), x0 + cdx(s))) {
      found = true;
      break;
    }
  } while (s != 4);
  if (!found) {  // single-pixel domain
    visit(x0, y0);
    return 0;
  }
  const int y1 = y0 + cdy(s), x1 = x0 + cdx(s);
  int y3 = y0, x3 = x0, prev_s = s ^ 4;
  long long area2 = 0;
  for (long long step = 0; step < max_steps; ++step) {
    int y4, x4;
    do {  // counter-clockwise search from the direction after the one we came from
      ++s;
      y4 = y3 + cdy(s & 7);
      x4 = x3 + cdx(s & 7);
    } while (!im.fg(y4, x4));
    s &= 7;
    if (s != prev_s) {
      visit(x3, y3);
      prev_s = s;
    }
    area2 += (long long)x3 * y4 - (long long)x4 * y3;
    if (y4 == y0 && x4 == x0 && y3 == y1 && x3 == x1) break;
    y3 = y4;
    x3 = x4;
    s = (s + 4) & 7;
  }
  return area2;
}

__global__ void __launch_bounds__(256) contour_init_kernel(const uint8_t* gray, const int* roi, int J, int H, int W,
                                                          int thresh, int* parent, unsigned long long* key, int* hist) {
  const long long n = (long long)J * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    const int job = (int)(i / ((long long)H * W));
    const int p = (int)(i - (long long)job * H * W);
    const ContourImg im = contour_img(gray, roi, job, H, W, thresh);
    parent[i] = im.fg(p / W, p % W) ? p : -1;
  }
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < J; i += gridDim.x * 256LL) key[i] = 0ull;
  if (hist)
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < (long long)J * 256; i += gridDim.x * 256LL) hist[i] = 0;
}

__device__ __forceinline__ int uf_find(int* parent, int x) {
  for (;;) {
    const int p = parent[x];
    if (p == x) return x;
    const int gp = parent[p];
    if (gp != p) parent[x] = gp;  // path halving (a benign race: every value written is an ancestor)
    x = p;
  }
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
    }
    if (atomicCAS(&parent[a], a, b) == a) return;  // the larger root hangs under the smaller one
  }
}

__global__ void __launch_bounds__(256) contour_union_kernel(int J, int H, int W, int* parent) {
  const long long n = (long long)J * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    if (parent[i] < 0) continue;
    const int job = (int)(i / ((long long)H * W));
    int* P = parent + (long long)job * H * W;
    const int p = (int)(i - (long long)job * H * W);
    const int y = p / W, x = p % W;
    // 8-connectivity through the four already-scanned neighbours
    if (x > 0 && P[p - 1] >= 0) uf_union(P, p, p - 1);
    if (y > 0) {
      if (P[p - W] >= 0) uf_union(P, p, p - W);
      if (x > 0 && P[p - W - 1] >= 0) uf_union(P, p, p - W - 1);
      if (x + 1 < W && P[p - W + 1] >= 0) uf_union(P, p, p - W + 1);
    }
  }
}

__global__ void __launch_bounds__(256) contour_trace_kernel(const uint8_t* gray, const int* roi, int J, int H, int W,
                                                           int thresh, const int* parent, unsigned long long* key) {
  const long long n = (long long)J * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    const int job = (int)(i / ((long long)H * W));
    const int p = (int)(i - (long long)job * H * W);
    if (parent[i] != p) continue;  // not the first raster pixel of a component
    const ContourImg im = contour_img(gray, roi, job, H, W, thresh);
    const long long a2 = trace_outer(im, p / W, p % W, 4LL * H * W + 16, [](int, int) {});
    const float area = 0.5f * (float)(a2 < 0 ? -a2 : a2);
    const unsigned long long k = ((unsigned long long)__float_as_uint(area) << 32) | (unsigned)(p + 1);
    atomicMax(&key[job], k);
  }
}

__global__ void __launch_bounds__(64) contour_finish_kernel(const uint8_t* gray, const int* roi, int J, int H, int W,
                                                           int thresh, const unsigned long long* key, int* mark,
                                                           const uint8_t* labels, const int* label_index, int* hist,
                                                           float* area_out, int* bbox_out, int* start_out,
                                                           int* vertex_count, int* label_out, int* label_count) {
  const int job = blockIdx.x * 64 + threadIdx.x;
  if (job >= J) return;
  const unsigned long long k = key[job];
  if (k == 0ull) {
    area_out[job] = 0.f;
    if (start_out) start_out[job] = -1;
    for (int c = 0; c < 4; ++c) bbox_out[4 * job + c] = 0;
    if (vertex_count) vertex_count[job] = 0;
    if (label_out) label_out[job] = 0;
    if (label_count) label_count[job] = 0;
    return;
  }
  const int p = (int)(unsigned)(k & 0xffffffffull) - 1;
  const ContourImg im = contour_img(gray, roi, job, H, W, thresh);
  int* M = mark + (long long)job * H * W;
  const uint8_t* L = labels ? labels + (long long)(label_index ? label_index[job] : job) * H * W : nullptr;
  int* hs = hist ? hist + 256 * job : nullptr;
  int xmin = W, xmax = -1, ymin = H, ymax = -1, nv = 0;
  trace_outer(im, p / W, p % W, 4LL * H * W + 16, [&](int x, int y) {
    xmin = min(xmin, x);
    xmax = max(xmax, x);
    ymin = min(ymin, y);
    ymax = max(ymax, y);
    const int q = y * W + x;
    if (M[q] != -7) {  // a pixel the border passes twice is one pixel of the drawn mask
      M[q] = -7;
      ++nv;
      if (hs) ++hs[L[q]];
    }
  });
  area_out[job] = __uint_as_float((unsigned)(k >> 32));
  if (start_out) start_out[job] = p;
  bbox_out[4 * job] = xmin;
  bbox_out[4 * job + 1] = ymin;
  bbox_out[4 * job + 2] = xmax - xmin + 1;
  bbox_out[4 * job + 3] = ymax - ymin + 1;
  if (vertex_count) vertex_count[job] = nv;
  if (hs) {  // sorted([(count, label)], reverse=True)[0]: the largest count, ties to the larger label
    int best = 0, bc = -1;
    for (int l = 0; l < 256; ++l)
      if (hs[l] > 0 && hs[l] >= bc) {
        bc = hs[l];
        best = l;
      }
    label_out[job] = best;
    label_count[job] = bc < 0 ? 0 : bc;
  }
}

}  // namespace cn

extern "C" size_t cn_contour_workspace_bytes(int32_t num_images, int32_t height, int32_t width) {
  return (size_t)num_images * height * width * sizeof(int32_t) + (size_t)num_images * (8 + 256 * sizeof(int32_t)) + 64;
}

extern "C" int cn_contour_largest(const uint8_t* gray, const int32_t* roi, int32_t num_images, int32_t height, int32_t width,
                                  int32_t thresh, const uint8_t* labels, const int32_t* label_index, float* area,
                                  int32_t* bbox, int32_t* start, int32_t* vertex_count, int32_t* label, int32_t* label_count,
                                  void* workspace, size_t workspace_bytes, cn_stream_t stream) {
  CN_REQUIRE(gray && area && bbox && workspace, CN_ERR_INVALID, "cn_contour_largest: null argument");
  CN_REQUIRE(num_images >= 0 && height > 0 && width > 0 && (long long)height * width < (1LL << 30), CN_ERR_INVALID,
             "cn_contour_largest: %d images of %d x %d", num_images, height, width);
  CN_REQUIRE(!labels || (vertex_count && label && label_count), CN_ERR_INVALID,
             "cn_contour_largest: vertex_count / label / label_count required with a label image");
  CN_REQUIRE(workspace_bytes >= cn_contour_workspace_bytes(num_images, height, width), CN_ERR_WORKSPACE,
             "cn_contour_largest: workspace %zu B < %zu B", workspace_bytes,
             cn_contour_workspace_bytes(num_images, height, width));
  CN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, CN_ERR_INVALID, "cn_contour_largest: workspace must be 8-B aligned");
  if (num_images == 0) return CN_OK;
  hipStream_t s = cn::as_stream(stream);
  const long long n = (long long)num_images * height * width;
  unsigned long long* key = static_cast<unsigned long long*>(workspace);
  int* hist = reinterpret_cast<int*>(key + num_images);
  int* parent = hist + (size_t)num_images * 256;
  const unsigned grid = cn::grid_for(n, 256, 1 << 16);
  hipLaunchKernelGGL(cn::contour_init_kernel, dim3(grid), dim3(256), 0, s, gray, roi, num_images, height, width, thresh,
                     parent, key, labels ? hist : nullptr);
  hipLaunchKernelGGL(cn::contour_union_kernel, dim3(grid), dim3(256), 0, s, num_images, height, width, parent);
  hipLaunchKernelGGL(cn::contour_trace_kernel, dim3(grid), dim3(256), 0, s, gray, roi, num_images, height, width, thresh,
                     parent, key);
  hipLaunchKernelGGL(cn::contour_finish_kernel, dim3((num_images + 63) / 64), dim3(64), 0, s, gray, roi, num_images, height,
                     width, thresh, key, parent, labels, label_index, labels ? hist : nullptr, area, bbox, start,
                     vertex_count, label, label_count);
  return cn::check_launch("cn_contour_largest");
}

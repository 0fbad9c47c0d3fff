// Batched semantic projection: the ray side of FruitModel.get_outputs_for_projections
// (crop_nerf/fruit_nerf/fruit_nerf.py:254-318) for MANY (camera, sub-cluster box) jobs in one launch sequence.
//
// The reference generates the rays of a whole frame per job (cam.generate_rays(aabb_box=...), :283), masks the ones that
// hit the box (:285), renders them, and writes two full-frame images (:299-315).  A boll-sized box covers a few thousand of
// a frame's 640 000 pixels, so per job almost all of that is fixed cost.  Here the host projects the 8 corners of every box
// and hands over, per job, the camera, the box and a screen RECTANGLE that contains every pixel whose ray can hit the box;
// the pixels of all rectangles of a batch are numbered consecutively ("slots"):
//   projection_test_kernel      slot -> (job, pixel) -> ray -> slab test: hit flag, job id, hit count per job
//   projection_finalize_kernel  jobs with fewer than `min_rays` hits are dropped whole (:293: "< 10 valid rays -> black")
//   [caller: list of the set flags = the batch's one host synchronisation]
//   projection_gather_kernel    hit list -> origins / directions / nears / fars / camera indices of ONE jagged ray bundle
//   [caller: sampler + render of the bundle (:301), density-only pass on (0, near) (:307-310)]
//   projection_scatter_kernel   semantics + occlusion weight per ray -> wo_occ / visible values per slot (:302, :311-313),
//                               as floats and / or as the uint8 a PNG round trip leaves
//   projection_paste_kernel     slot values -> full-frame uint8 images (for the in-process merger)
// The ray arithmetic is raygen_dev.hpp's, i.e. the bits of cn_raygen_pinhole + cn_intersect_aabb for the same pixel.
#include "cn_common.hpp"
#include "raygen_dev.hpp"

namespace cn {

// job of a slot: the last job whose slot_offset <= slot (jobs with empty rectangles share an offset with their successor)
__device__ __forceinline__ int job_of(const cn_projection_job* __restrict__ jobs, int num_jobs, long long slot) {
  int lo = 0, hi = num_jobs - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].slot_offset <= slot) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

struct SlotRay {
  float ox, oy, oz, dx, dy, dz, near, far;
  int pixel;  // row * image_width + col
};

__device__ __forceinline__ SlotRay slot_ray(const cn_projection_job& J, long long local, int image_width) {
  const int ry = (int)(local / J.w), rx = (int)(local % J.w);
  const long long row = J.y0 + ry, col = J.x0 + rx;
  float cx0, cy0, n0;
  pixel_camera_coords(J.fx, J.fy, J.cx, J.cy, row, col, cx0, cy0);
  SlotRay r;
  rotate_normalize(J.c2w, cx0, cy0, r.dx, r.dy, r.dz, n0);
  r.ox = J.c2w[3];
  r.oy = J.c2w[7];
  r.oz = J.c2w[11];
  slab_test(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, J.aabb[0], J.aabb[1], J.aabb[2], J.aabb[3], J.aabb[4], J.aabb[5], r.near,
            r.far);
  r.pixel = (int)(row * image_width + col);
  return r;
}

__global__ void __launch_bounds__(256)
projection_test_kernel(const cn_projection_job* __restrict__ jobs, int num_jobs, long long num_slots, int image_width,
                       uint8_t* __restrict__ flags, int* __restrict__ job_of_slot, int* __restrict__ hit_count) {
  const long long slot = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const bool live = slot < num_slots;
  int j = -1;
  bool hit = false;
  if (live) {
    j = job_of(jobs, num_jobs, slot);
    const cn_projection_job J = jobs[j];
    hit = slot_ray(J, slot - J.slot_offset, image_width).near < 1e10f;  // valid_rays_mask = rays.nears < 1e10 (:285)
    flags[slot] = hit ? 1 : 0;
    job_of_slot[slot] = j;
  }
  // hit count per job: one atomic per wave when the whole wave works on one job (the usual case), else one per hit lane
  const unsigned long long hits = __ballot(hit);
  const int j0 = __builtin_amdgcn_readfirstlane(j);
  const bool uniform = __ballot(live && j != j0) == 0ull;
  if (uniform) {
    if (hits && (__lane_id() == (unsigned)__builtin_ctzll(__ballot(live)))) atomicAdd(&hit_count[j0], __popcll(hits));
  } else if (hit) {
    atomicAdd(&hit_count[j], 1);
  }
}

__global__ void __launch_bounds__(256)
projection_finalize_kernel(long long num_slots, int min_rays, const int* __restrict__ job_of_slot,
                           const int* __restrict__ hit_count, uint8_t* __restrict__ flags) {
  const long long slot = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (slot >= num_slots) return;
  if (flags[slot] && hit_count[job_of_slot[slot]] < min_rays) flags[slot] = 0;
}

__global__ void __launch_bounds__(256)
projection_gather_kernel(const cn_projection_job* __restrict__ jobs, const int* __restrict__ job_of_slot,
                         const int64_t* __restrict__ hit_slots, long long num_hits, int image_width,
                         float* __restrict__ origins, float* __restrict__ directions, float* __restrict__ nears,
                         float* __restrict__ fars, int64_t* __restrict__ camera_indices, int* __restrict__ ray_job,
                         int* __restrict__ ray_pixel) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= num_hits) return;
  const long long slot = hit_slots[i];
  const int j = job_of_slot[slot];
  const cn_projection_job J = jobs[j];
  const SlotRay r = slot_ray(J, slot - J.slot_offset, image_width);
  origins[3 * i + 0] = r.ox;
  origins[3 * i + 1] = r.oy;
  origins[3 * i + 2] = r.oz;
  directions[3 * i + 0] = r.dx;
  directions[3 * i + 1] = r.dy;
  directions[3 * i + 2] = r.dz;
  nears[i] = r.near;
  fars[i] = r.far;
  camera_indices[i] = (int64_t)J.camera_index;
  if (ray_job) ray_job[i] = j;
  if (ray_pixel) ray_pixel[i] = r.pixel;
}

// torchvision.utils.save_image on a float image: clamp(0, 1) * 255 + 0.5, clamp(0, 255), to uint8 -- two roundings (a
// multiply, then an add: no fused multiply-add), truncation
__device__ __forceinline__ uint8_t quantise_unit(float v) {
  float c = fminf(fmaxf(v, 0.f), 1.f);
  float q = __fadd_rn(__fmul_rn(c, 255.f), 0.5f);
  q = fminf(fmaxf(q, 0.f), 255.f);
  return (uint8_t)(int)q;
}

__global__ void __launch_bounds__(256)
projection_scatter_kernel(const float* __restrict__ semantics, const float* __restrict__ occlusion,
                          const int64_t* __restrict__ hit_slots, long long num_hits, float occlusion_threshold,
                          float* __restrict__ wo_occ_f32, float* __restrict__ visible_f32,
                          uint8_t* __restrict__ wo_occ_u8, uint8_t* __restrict__ visible_u8) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= num_hits) return;
  const long long slot = hit_slots[i];
  const float v = semantics[i];                              // rgb_img[valid] = outputs['semantics'] (:302)
  const bool hidden = occlusion[i] >= occlusion_threshold;   // occlusion_mark = weights >= .5 (:311)
  const float vis = hidden ? 0.f : v;                        // rgb_img[occlusion_mark] = 0.0 (:313)
  if (wo_occ_f32) wo_occ_f32[slot] = v;
  if (visible_f32) visible_f32[slot] = vis;
  if (wo_occ_u8) wo_occ_u8[slot] = quantise_unit(v);
  if (visible_u8) visible_u8[slot] = quantise_unit(vis);
}

__global__ void __launch_bounds__(256)
projection_paste_kernel(const cn_projection_job* __restrict__ jobs, const int* __restrict__ job_of_slot,
                        const uint8_t* __restrict__ slot_values, long long num_slots, const int* __restrict__ image_of_job,
                        int num_images, int image_height, int image_width, uint8_t* __restrict__ images) {
  const long long slot = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (slot >= num_slots) return;
  const int j = job_of_slot[slot];
  const int img = image_of_job[j];
  if (img < 0 || img >= num_images) return;
  const long long local = slot - jobs[j].slot_offset;
  const int w = jobs[j].w;
  const long long row = jobs[j].y0 + local / w, col = jobs[j].x0 + local % w;
  images[((long long)img * image_height + row) * image_width + col] = slot_values[slot];
}

}  // namespace cn

extern "C" int cn_projection_test(const cn_projection_job* jobs, int32_t num_jobs, int64_t num_slots, int32_t image_width,
                                  int32_t min_rays, uint8_t* flags, int32_t* job_of_slot, int32_t* hit_count,
                                  cn_stream_t stream) {
  CN_REQUIRE(num_jobs >= 0 && num_slots >= 0, CN_ERR_INVALID, "cn_projection_test: negative size");
  if (num_jobs == 0 || num_slots == 0) return CN_OK;
  CN_REQUIRE(jobs && flags && job_of_slot && hit_count, CN_ERR_INVALID, "cn_projection_test: null argument");
  CN_REQUIRE(image_width > 0, CN_ERR_INVALID, "cn_projection_test: image_width %d", image_width);
  CN_REQUIRE(num_slots < (1LL << 31), CN_ERR_INVALID, "cn_projection_test: at most 2^31-1 slots per batch");
  hipStream_t s = cn::as_stream(stream);
  hipError_t e = hipMemsetAsync(hit_count, 0, sizeof(int32_t) * (size_t)num_jobs, s);
  CN_REQUIRE(e == hipSuccess, CN_ERR_LAUNCH, "cn_projection_test: %s", hipGetErrorString(e));
  const unsigned blocks = (unsigned)((num_slots + 255) / 256);
  hipLaunchKernelGGL(cn::projection_test_kernel, dim3(blocks), dim3(256), 0, s, jobs, num_jobs, (long long)num_slots,
                     image_width, flags, job_of_slot, hit_count);
  int rc = cn::check_launch("cn_projection_test");
  if (rc) return rc;
  if (min_rays > 0) {
    hipLaunchKernelGGL(cn::projection_finalize_kernel, dim3(blocks), dim3(256), 0, s, (long long)num_slots, min_rays,
                       job_of_slot, hit_count, flags);
    rc = cn::check_launch("cn_projection_test (finalize)");
  }
  return rc;
}

extern "C" int cn_projection_gather(const cn_projection_job* jobs, const int32_t* job_of_slot, const int64_t* hit_slots,
                                    int64_t num_hits, int32_t image_width, float* origins, float* directions, float* nears,
                                    float* fars, int64_t* camera_indices, int32_t* ray_job, int32_t* ray_pixel,
                                    cn_stream_t stream) {
  if (num_hits <= 0) return CN_OK;
  CN_REQUIRE(jobs && job_of_slot && hit_slots && origins && directions && nears && fars && camera_indices, CN_ERR_INVALID,
             "cn_projection_gather: null argument");
  CN_REQUIRE(image_width > 0, CN_ERR_INVALID, "cn_projection_gather: image_width %d", image_width);
  hipLaunchKernelGGL(cn::projection_gather_kernel, dim3((unsigned)((num_hits + 255) / 256)), dim3(256), 0,
                     cn::as_stream(stream), jobs, job_of_slot, hit_slots, (long long)num_hits, image_width, origins,
                     directions, nears, fars, camera_indices, ray_job, ray_pixel);
  return cn::check_launch("cn_projection_gather");
}

extern "C" int cn_projection_scatter(const float* semantics, const float* occlusion, const int64_t* hit_slots,
                                     int64_t num_hits, float occlusion_threshold, float* wo_occ_f32, float* visible_f32,
                                     uint8_t* wo_occ_u8, uint8_t* visible_u8, cn_stream_t stream) {
  if (num_hits <= 0) return CN_OK;
  CN_REQUIRE(semantics && occlusion && hit_slots, CN_ERR_INVALID, "cn_projection_scatter: null argument");
  hipLaunchKernelGGL(cn::projection_scatter_kernel, dim3((unsigned)((num_hits + 255) / 256)), dim3(256), 0,
                     cn::as_stream(stream), semantics, occlusion, hit_slots, (long long)num_hits, occlusion_threshold,
                     wo_occ_f32, visible_f32, wo_occ_u8, visible_u8);
  return cn::check_launch("cn_projection_scatter");
}

extern "C" int cn_projection_paste(const cn_projection_job* jobs, const int32_t* job_of_slot, const uint8_t* slot_values,
                                   int64_t num_slots, const int32_t* image_of_job, int32_t num_images,
                                   int32_t image_height, int32_t image_width, uint8_t* images, cn_stream_t stream) {
  if (num_slots <= 0) return CN_OK;
  CN_REQUIRE(jobs && job_of_slot && slot_values && image_of_job && images, CN_ERR_INVALID,
             "cn_projection_paste: null argument");
  CN_REQUIRE(num_images > 0 && image_height > 0 && image_width > 0, CN_ERR_INVALID,
             "cn_projection_paste: %d images of %d x %d", num_images, image_height, image_width);
  hipLaunchKernelGGL(cn::projection_paste_kernel, dim3((unsigned)((num_slots + 255) / 256)), dim3(256), 0,
                     cn::as_stream(stream), jobs, job_of_slot, slot_values, (long long)num_slots, image_of_job,
                     num_images, image_height, image_width, images);
  return cn::check_launch("cn_projection_paste");
}

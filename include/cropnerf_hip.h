/*
 * cropnerf_hip.h -- C ABI of libcropnerf_hip.so (gfx950 / MI355X)
 *
 * Drop-in boundary for the volumetric ray-marching hot path of CropNeRF's `fruit_nerf`
 * method.  The reference (pure Python, /root/reference/crop_nerf) has no FFI of its own:
 * every entry point below replaces a Python call site of the reference (cited per
 * function as file:line under crop_nerf/) whose arithmetic lives in nerfstudio 1.1.3.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (CN_OK) or a negative cn_status; cn_last_error() gives a
 *     thread-local message for the last failure on the calling thread.
 *   - all array arguments are DEVICE pointers unless marked "host"; fp32 unless noted;
 *     indices int64 (nerfstudio uses torch.long).  The caller owns every buffer; the
 *     library never allocates device memory -- scratch comes through `workspace`,
 *     sized by the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised.
 *   - re-entrant; no global mutable state.  One process per GPU for multi-GPU use.
 *   - layouts: rays SoA [R,3]/[R,1] row-major; samples [R,S]; weights of Linear layers in
 *     torch.nn.Linear layout [out,in]; hash tables [entries, 2] in one of the two layouts of cn_grid (nerfstudio torch
 *     HashEncoding: [num_levels * 2^log2_T, 2] level-major; or the tcnn-compatible layout of cn_tcnn_grid_plan).
 */
#ifndef CROPNERF_HIP_H
#define CROPNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cn_status {
  CN_OK = 0,
  CN_ERR_INVALID = -1,      /* bad argument (null pointer, size, enum)            */
  CN_ERR_UNSUPPORTED = -2,  /* dimensions outside what the kernels are built for  */
  CN_ERR_LAUNCH = -3,       /* HIP launch / runtime error                          */
  CN_ERR_WORKSPACE = -4     /* workspace too small                                 */
} cn_status;

typedef void* cn_stream_t; /* hipStream_t */

#define CN_MAX_LEVELS 16
#define CN_MAX_LAYERS 4

/* spacing functions of the SpacedSampler family (fruit_nerf/components/ray_samplers.py:46-52) */
#define CN_SPACING_UNIFORM 0   /* s(x)=x                                                       */
#define CN_SPACING_PIECEWISE 1 /* s(x)=x/2 (x<1) else 1-1/(2x): UniformLinDispPiecewiseSampler */

/* appearance-embedding modes of FruitField (fruit_nerf/fruit_field.py:218-220,250-261) */
#define CN_APP_ZEROS 0
#define CN_APP_MEAN 1
#define CN_APP_PER_CAMERA 2

/* background modes of RGBRenderer (fruit_nerf/fruit_nerf.py:170; scripts/semantic_projection.py:158,169) */
#define CN_MATRIX_FP32 0
#define CN_MATRIX_SPLIT_BF16 1
#define CN_MATRIX_F16 2

#define CN_BG_LAST_SAMPLE 0
#define CN_BG_COLOR 1

/* Hash-grid layouts.  The reference builds FruitField with implementation="tcnn" by default (fruit_nerf/fruit_field.py:95,
 * 125-132) and falls back to nerfstudio's torch HashEncoding with implementation="torch"; the two differ in indexing.
 *   CN_GRID_TORCH  nerfstudio torch HashEncoding: every level hashed into 2^log2_table_size entries, level l at entry
 *                  l << log2_table_size, cell = floor(x * scalings[l]), scalings = floor(min_res * growth^l).
 *   CN_GRID_TCNN   tiny-cuda-nn GridEncoding (encodings/grid.h): cell = floor(x * scalings[l] + 0.5) with
 *                  scalings[l] = exp2(l * log2(per_level_scale)) * base_resolution - 1; a level whose resolution^3 fits the
 *                  table is DENSE, the others are hashed (same primes).  This library keeps a dense level with
 *                  power-of-two strides -- entry = level_offset[l] + (x | y << b | z << 2b), b = level_bits[l] -- so that one
 *                  index expression (xor of three per-axis terms) serves hashed and dense levels alike; cn_tcnn_grid_pack /
 *                  cn_tcnn_grid_unpack convert from / to tcnn's packed parameter vector (x + y*res + z*res^2, levels
 *                  back to back), including the wrap-around entries tcnn reads when a corner index reaches `res`. */
#define CN_GRID_TORCH 0
#define CN_GRID_TCNN 1
/* element type of a hash table: two features per entry, as float2 (8 B) or half2 (4 B, what tcnn computes with) */
#define CN_TABLE_F32 0
#define CN_TABLE_F16 1

/* One multiresolution hash grid (fruit_nerf/fruit_field.py:125-132; proposal nets fruit_nerf/fruit_nerf.py:133-142).
 * A zero-initialised tail (layout .. scatter_scratch_bytes) is the torch layout with an fp32 table and no scratch. */
typedef struct cn_grid {
  const void* table;              /* [entries, 2] float (CN_TABLE_F32) or _Float16 (CN_TABLE_F16)                */
  int32_t num_levels;             /* <= CN_MAX_LEVELS                                                             */
  int32_t log2_table_size;        /* 2^k entries per hashed level                                                 */
  float scalings[CN_MAX_LEVELS];  /* host values, see the layouts above                                           */
  int32_t layout;                 /* CN_GRID_*                                                                    */
  int32_t table_dtype;            /* CN_TABLE_*                                                                   */
  uint32_t level_offset[CN_MAX_LEVELS]; /* CN_GRID_TCNN: first entry of level l                                  */
  uint8_t level_bits[CN_MAX_LEVELS];    /* CN_GRID_TCNN: 0 = hashed level, b > 0 = dense level with b bits per axis */
  /* Only read when the grid is a GRADIENT target (the `grads` argument of the backward entry points): optional scratch
   * of cn_grid_scatter_scratch_bytes(grid) bytes, zeroed ONCE by the caller; every backward call leaves it zeroed.  With
   * it the gradients of the coarse levels do not go to `table` sample by sample: cell-major records for the levels that hold
   * fewer cells than half the batch has samples (all three backward entry points; 64 private dense copies of level 0's
   * vertices when that is no level at all): a sample adds the 16 weighted values of its cell -- 8 corners
   * x 2 features -- to the cell's 64-byte record in ONE atomic request (the hash table takes one per x-edge, 4.5 per
   * sample and level), consecutive samples in one cell merge, and a fold kernel launched by the same call adds the
   * touched records to `table` and zeroes them.  NULL: every level goes straight to `table`. */
  void* scatter_scratch;
  uint64_t scatter_scratch_bytes;
} cn_grid;
size_t cn_grid_scatter_scratch_bytes(const cn_grid* grid);
/* The same for backward calls of at most max_samples samples: only the levels that such a call can select (at most
 * 2 x max_samples cells) -- the scratch may be any prefix of the full layout; <= 0: every level. */
size_t cn_grid_scatter_scratch_bytes_for(const cn_grid* grid, int64_t max_samples);

/* tiny-cuda-nn grid geometry for (n_levels, log2_hashmap_size, base_resolution, per_level_scale) -- what nerfstudio's
 * HashEncoding(implementation="tcnn") passes to tcnn.Encoding -- and this library's table layout for it. host struct. */
typedef struct cn_tcnn_grid_plan {
  int32_t num_levels;
  int32_t log2_table_size;
  int32_t base_resolution;
  float per_level_scale;
  float scalings[CN_MAX_LEVELS];               /* grid.h grid_scale(l)                                  */
  uint32_t resolution[CN_MAX_LEVELS];          /* grid.h grid_resolution(scale) = ceil(scale) + 1       */
  uint32_t packed_offset[CN_MAX_LEVELS + 1];   /* tcnn's offset table, in entries                       */
  uint32_t level_offset[CN_MAX_LEVELS + 1];    /* this library's table, in entries; [num_levels] = total */
  uint8_t level_bits[CN_MAX_LEVELS];
} cn_tcnn_grid_plan;

/* A nerfstudio torch MLP: Linear(+ReLU) x (num_layers-1), Linear. dims[0]=in, dims[num_layers]=out. */
typedef struct cn_mlp {
  int32_t num_layers;
  int32_t dims[CN_MAX_LAYERS + 1];
  const float* weight[CN_MAX_LAYERS]; /* [dims[i+1], dims[i]] */
  const float* bias[CN_MAX_LAYERS];   /* [dims[i+1]]          */
} cn_mlp;

/* FruitField parameters (module graph fruit_nerf/fruit_field.py:109-167). */
typedef struct cn_field_params {
  cn_grid grid;                 /* mlp_base_grid: 16 levels, F=2                               */
  cn_mlp base;                  /* mlp_base_mlp: 32 -> 64 -> 1+geo                             */
  cn_mlp semantics;             /* mlp_semantics: geo -> Hs (-> Hs) -> Ht, no out activation  */
  const float* sem_head_weight; /* field_head_semantics: [1, Ht]                               */
  const float* sem_head_bias;   /* [1]                                                         */
  cn_mlp color;                 /* mlp_head: 16+geo+app -> 64 -> 64 -> 3, sigmoid              */
  const float* appearance;      /* embedding_appearance: [num_images, app_dim]                */
  int32_t num_images;
  int32_t app_dim;
  int32_t geo_feat_dim;
} cn_field_params;

/* A proposal network: HashMLPDensityField (fruit_nerf/fruit_nerf.py:133-142). */
typedef struct cn_density_params {
  cn_grid grid; /* 5 levels (7 in fruit_nerf_method_huge) */
  cn_mlp mlp;   /* 2L -> 16 -> 1                            */
} cn_density_params;

/* Scene frame: SceneBox aabb + whether SceneContraction(inf) is active
 * (fruit_nerf/fruit_nerf.py:91-94,189; fruit_nerf/fruit_field.py:171-176). host struct. */
typedef struct cn_scene {
  float aabb[6];       /* min xyz, max xyz */
  int32_t contraction; /* 1: contract then (p+2)/4; 0: AABB-normalise */
} cn_scene;

/* Options of the fused renderer. host struct. */
typedef struct cn_render_opts {
  int32_t num_samples;  /* S: field samples per ray                                          */
  int32_t spacing;      /* CN_SPACING_* used when `bins` is NULL                             */
  int32_t bg_mode;      /* CN_BG_*                                                            */
  float bg_color[3];    /* used with CN_BG_COLOR                                              */
  int32_t app_mode;     /* CN_APP_*                                                           */
  int32_t sh_unit_dir;  /* 1: SH of the unit direction (tcnn semantics); 0: SH of (d+1)/2   */
  int32_t eval_clamp;   /* 1: nan_to_num(rgb) before, clamp[0,1] after (RGBRenderer in eval) */
  int32_t density_only; /* 1: only accumulation is produced (get_density_for_camera_ray_bundle) */
  /* Scheduling hint, never changes results: when image_width > 0 the rays of the call are pixels
   * [pixel_start, pixel_start + num_rays) of a row-major image of that width (the chunks of
   * get_outputs_for_camera_ray_bundle, fruit_nerf.py:388-391); the kernel then gives each XCD a column
   * stripe instead of a run of rows, so that vertically adjacent pixels share an L2. 0 = unknown order. */
  int32_t image_width;
  int64_t pixel_start;
  /* Early ray termination, an extension (the reference composites every sample): > 0 lets cn_render_rays
   * stop a ray after a 64-sample chunk once the transmittance behind it is below this value; every output
   * then differs from the full result by less than the threshold (times the value range).  0 = off. */
  float early_stop_transmittance;
  /* Arithmetic of the MLP matrix products, an extension.  CN_MATRIX_FP32 (0, default): exact fp32 products
   * (v_mfma_f32_16x16x4_f32).  CN_MATRIX_SPLIT_BF16 (1): every operand split into bf16 hi + lo, a.b ~ a_hi.b_hi +
   * a_hi.b_lo + a_lo.b_hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- about 16 mantissa bits per product,
   * i.e. between the reference's two own precisions (fp16 with tcnn, fp32 with torch), at a third of the matrix time.
   * Honoured by the producer/consumer render kernel (batches that fill the device); elsewhere products stay fp32.
   * CN_MATRIX_F16 (2): the reference's OWN arithmetic class -- tiny-cuda-nn's FullyFusedMLP under mixed precision
   * (fruit_nerf/fruit_field.py:95,125-167 build every module with implementation="tcnn"; fruit_nerf_config.py:35
   * mixed_precision=True): weights and layer inputs rounded to fp16 (a no-op for the weights of an imported tcnn
   * checkpoint, which are fp16 values already), products on v_mfma_f32_16x16x32_f16 with fp32 accumulation (tcnn
   * accumulates in fp16: this is at least as precise), the hash-grid interpolation of a half table carried out on
   * packed fp16 pairs as tcnn's kernel_grid does.  cn_render_rays / cn_render_samples then run the same kernel
   * whatever the batch size, so that a ray's result does not depend on the call it is part of. */
  int32_t matrix_precision;
} cn_render_opts;

const char* cn_last_error(void);
int cn_version(void);

/* ---------------------------------------------------------------------------------------------
 * tcnn-compatible hash grids: what a reference-trained checkpoint holds (FruitField's default implementation="tcnn",
 * fruit_nerf/fruit_field.py:95,125-132; loaded by scripts/exporter.py:87, scripts/semantic_projection.py:139-143).
 * ------------------------------------------------------------------------------------------- */

/* host only: fill `plan` for a tcnn HashGrid encoding (linear interpolation, 2 features per level, 3-D input). */
int cn_tcnn_grid_plan_init(int32_t num_levels, int32_t log2_table_size, int32_t base_resolution, float per_level_scale,
                           cn_tcnn_grid_plan* plan);

/* host only: the cn_grid that describes `table` (plan->level_offset[num_levels] entries of table_dtype). */
int cn_tcnn_grid_describe(const cn_tcnn_grid_plan* plan, const void* table, int32_t table_dtype, cn_grid* grid);

/* tcnn parameter vector (packed_offset[num_levels] entries x 2 values, dtype packed_dtype: the fp32 master copy of a
 * checkpoint or its fp16 cast) -> this library's table; every entry of `table` is written (unreachable ones with 0). */
int cn_tcnn_grid_pack(const cn_tcnn_grid_plan* plan, const void* packed, int32_t packed_dtype, void* table,
                      int32_t table_dtype, cn_stream_t stream);

/* the inverse: `packed` receives the value of every tcnn parameter (the padding entries of dense levels included). */
int cn_tcnn_grid_unpack(const cn_tcnn_grid_plan* plan, const void* table, int32_t table_dtype, void* packed,
                        int32_t packed_dtype, cn_stream_t stream);

/* Training on a tcnn-layout table: a corner index that reaches `res` on a dense level wraps into a neighbouring row of
 * tcnn's x + y*res + z*res^2 array, i.e. several entries of this library's table stand for ONE tcnn parameter.
 * cn_tcnn_grid_tie_gradients folds the gradient of every such alias into its parameter's own entry (and zeroes the
 * alias); cn_tcnn_grid_tie_parameters copies the parameter back into its aliases after the optimiser step. fp32 tables. */
int cn_tcnn_grid_tie_gradients(const cn_tcnn_grid_plan* plan, float* grad_table, cn_stream_t stream);
int cn_tcnn_grid_tie_parameters(const cn_tcnn_grid_plan* plan, float* table, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Ray generation
 * ------------------------------------------------------------------------------------------- */

/* Pinhole rays.  Replaces Cameras.generate_rays as called at
 *   fruit_nerf/data/fruit_datamanager.py:188-197 (train_ray_generator(ray_indices)),
 *   fruit_nerf/fruit_nerf.py:283 (full image, camera_indices=0),
 *   fruit_nerf/export/exporter_utils_nerfacto.py:266-268.
 * ray_indices [R,3] = (camera, row, col) or NULL; with NULL the rays are pixels
 * [pixel_start, pixel_start+R) of camera `cam`, row-major over (H,W).
 * camera_index_value >= 0 overrides the camera index written to `camera_indices`
 * (the reference writes 0 at fruit_nerf.py:283). Outputs may be NULL to skip. */
int cn_raygen_pinhole(const float* c2w /*[C,3,4]*/, const float* intrinsics /*[C,4] fx fy cx cy*/,
                      const int64_t* ray_indices, int32_t cam, int32_t height, int32_t width,
                      int64_t pixel_start, int64_t num_rays, int32_t camera_index_value,
                      float* origins, float* directions, float* pixel_area, int64_t* camera_indices,
                      float* directions_norm, cn_stream_t stream);

/* Ray / AABB slab test: nerfstudio.utils.math.intersect_aabb as used by generate_rays(aabb_box=...)
 * at fruit_nerf/fruit_nerf.py:283-286 (misses -> 1e10 in both outputs). aabb: host [6]. */
int cn_intersect_aabb(const float* origins, const float* directions, const float* aabb_host,
                      int64_t num_rays, float* nears, float* fars, cn_stream_t stream);

/* Orthographic surface rays: OrthographicRayGenerator.forward
 * (fruit_nerf/components/ray_generators.py:46-66) over the surface grid of
 * fruit_nerf/data/fruit_datamanager.py:71-121.  surface_points [P,3] device; rays
 * [start, start+num_rays). plane_vector: host [3]. */
int cn_raygen_ortho(const float* surface_points, const float* plane_vector_host, int64_t start,
                    int64_t num_rays, float* origins, float* directions, float* pixel_area, float* nears,
                    float* fars, cn_stream_t stream);

/* Dense n x n surface grid (sample_surface_points, fruit_nerf/data/fruit_datamanager.py:71-121):
 * point i*ny+j = (lerp x, lerp y, z_const). */
int cn_surface_grid(float x0, float x1, int32_t nx, float y0, float y1, int32_t ny, float z_const,
                    float* surface_points, cn_stream_t stream);

/* SO3xR3 pose refinement: camera_optimizer.apply_to_raybundle (fruit_nerf/fruit_nerf.py:547).
 * origins/directions are updated in place. */
int cn_apply_pose_adjustment(const float* pose_adjustment /*[C,6]*/, const int64_t* camera_indices,
                             int64_t num_rays, float* origins, float* directions, cn_stream_t stream);
/* The same out of place (the training step keeps the raw rays for the pose gradient: no clone before the tweak). */
int cn_apply_pose_adjustment_to(const float* pose_adjustment, const int64_t* camera_indices, int64_t num_rays,
                                const float* origins, const float* directions, float* out_origins,
                                float* out_directions, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Batched semantic projection (FruitModel.get_outputs_for_projections, fruit_nerf/fruit_nerf.py:254-318)
 *
 * The reference loops over (super-cluster, camera, sub-cluster box) jobs and, per job, generates the rays of a whole
 * frame with the box's slab test (:283), renders the rays that hit (:299-301), renders (0, near) of the same rays for the
 * occlusion weight (:305-310) and writes two full-frame images (:302-304, :311-315).  These entry points do the ray side
 * of that for a BATCH of jobs: per job the caller gives the camera, the box and a screen rectangle that contains every
 * pixel whose ray can hit the box (projected box corners; the whole frame is always a valid rectangle).  The pixels of all
 * rectangles, job after job and row-major inside a rectangle, are the batch's "slots"; slot_offset = a job's first slot.
 * ------------------------------------------------------------------------------------------- */

/* One projection job.  DEVICE array (the kernels read it); the host fills it and copies it over. */
typedef struct cn_projection_job {
  float c2w[12];        /* camera-to-world, 3x4 row-major                                           */
  float fx;             /* intrinsics of that camera                                                */
  float fy;
  float cx;
  float cy;
  float aabb[6];        /* the sub-cluster box: min xyz, max xyz                                    */
  int32_t x0;           /* screen rectangle: first column ...                                       */
  int32_t y0;           /* ... first row ...                                                        */
  int32_t w;            /* ... width and ...                                                        */
  int32_t h;            /* ... height in pixels (inside the frame; may be empty)                    */
  int32_t camera_index; /* value written to camera_indices (the reference writes 0, :283)           */
  int32_t reserved;
  int64_t slot_offset;  /* sum of w*h of the jobs before this one                                   */
} cn_projection_job;

/* Step 1.  Per slot: the ray of its pixel (bit for bit cn_raygen_pinhole's) and the slab test against its job's box
 * (cn_intersect_aabb's): flags[slot] = 1 when it hits (valid_rays_mask, :285), job_of_slot[slot], hit_count[job]
 * (zeroed here).  Then every flag of a job with fewer than min_rays hits is cleared (:293 writes black images for
 * such a job).  The caller lists the set flags -- its one synchronisation per batch. */
int cn_projection_test(const cn_projection_job* jobs, int32_t num_jobs, int64_t num_slots, int32_t image_width,
                       int32_t min_rays, uint8_t* flags /*[P]*/, int32_t* job_of_slot /*[P]*/,
                       int32_t* hit_count /*[num_jobs]*/, cn_stream_t stream);

/* Step 2.  hit_slots [N] (ascending slot numbers of the set flags) -> ONE jagged ray bundle: origins, directions [N,3],
 * nears, fars [N] from the slab test, camera_indices [N] = the job's camera_index; optionally the job and the pixel
 * (row * image_width + col) of every ray. */
int cn_projection_gather(const cn_projection_job* jobs, const int32_t* job_of_slot, const int64_t* hit_slots,
                         int64_t num_hits, int32_t image_width, float* origins, float* directions, float* nears,
                         float* fars, int64_t* camera_indices, int32_t* ray_job /*or NULL*/,
                         int32_t* ray_pixel /*or NULL*/, cn_stream_t stream);

/* Step 3, after the two renders of the bundle.  semantics [N] = outputs['semantics'] of the box-restricted render,
 * occlusion [N] = sum of the weights on (0, near).  Per ray, into its slot: wo_occ = semantics (:302), visible =
 * semantics where occlusion < occlusion_threshold, else 0 (:311-313).  As floats and / or as the uint8 that
 * torchvision.utils.save_image writes (clamp(0,1) * 255 + 0.5, truncated).  Outputs may be NULL; the caller zeroes them
 * first (slots without a hit stay 0 = the black of the reference's images). */
int cn_projection_scatter(const float* semantics, const float* occlusion, const int64_t* hit_slots, int64_t num_hits,
                          float occlusion_threshold, float* wo_occ_f32, float* visible_f32, uint8_t* wo_occ_u8,
                          uint8_t* visible_u8, cn_stream_t stream);

/* Slot values -> full frames: images[image_of_job[j]] ([num_images, H, W] uint8, zeroed by the caller) receives job j's
 * rectangle (an index outside [0, num_images): skipped).  For consumers that want whole images on the device (the merger's contour
 * stage); the PNG writer assembles its frames on the host from the slot values instead. */
int cn_projection_paste(const cn_projection_job* jobs, const int32_t* job_of_slot, const uint8_t* slot_values,
                        int64_t num_slots, const int32_t* image_of_job, int32_t num_images, int32_t image_height,
                        int32_t image_width, uint8_t* images, cn_stream_t stream);

/* The projection stage's files: for each of `count` jobs an 8-bit RGB PNG of an image_height x image_width frame that is
 * black except for the gray rectangle rects[i] = (x0, y0, w, h) filled, row-major, from values[value_offsets[i] ...] (HOST
 * memory: the slot values of cn_projection_scatter copied back) -- the file torchvision.utils.save_image writes at
 * fruit_nerf/fruit_nerf.py:304,315 up to the compressed byte stream (same decoded pixels).  make_dirs != 0 creates missing
 * parent directories.  Pure host code, thread-safe, no stream: meant to be called from worker threads behind the GPU. */
int cn_png_write_gray_rects(int32_t count, const char* const* paths, const uint8_t* values, const int64_t* value_offsets,
                            const int32_t* rects /*[count,4]*/, int32_t image_height, int32_t image_width,
                            int32_t make_dirs);

/* ---------------------------------------------------------------------------------------------
 * Samplers
 * ------------------------------------------------------------------------------------------- */

/* SpacedSampler / UniformSamplerWithNoise.generate_ray_samples
 * (fruit_nerf/components/ray_samplers.py:54-104).  t_rand: NULL (eval) or stratified jitter with row
 * stride t_rand_stride (1 = single jitter, S+1 = per-bin).  Outputs [R,S]; any may be NULL. */
int cn_sample_spaced(const float* nears, const float* fars, int64_t num_rays, int32_t num_samples,
                     int32_t spacing, const float* t_rand, int32_t t_rand_stride, float* starts,
                     float* ends, float* spacing_starts, float* spacing_ends, cn_stream_t stream);

/* PDFSampler.generate_ray_samples (nerfstudio; called through ProposalNetworkSampler at
 * fruit_nerf/fruit_nerf.py:157-164,549).  Inputs: previous spacing bins [R,S_in+1] and weights [R,S_in]
 * (already annealed by the caller or via `anneal`: w^anneal).  u_rand NULL = eval.
 * Outputs: spacing bins [R,S_out+1] and euclidean bins [R,S_out+1]. */
int cn_sample_pdf(const float* prev_spacing_bins, const float* weights, const float* nears, const float* fars,
                  int64_t num_rays, int32_t s_in, int32_t s_out, float anneal, int32_t spacing,
                  const float* u_rand, int32_t u_rand_stride, float* spacing_bins, float* euclidean_bins,
                  cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Field evaluation (materialised, per-sample outputs)
 * ------------------------------------------------------------------------------------------- */

/* HashMLPDensityField.density_fn at sample mid-points (fruit_nerf/fruit_nerf.py:135-142). density [R,S]. */
int cn_proposal_density(const cn_density_params* params, const cn_scene* scene, const float* origins,
                        const float* directions, const float* starts, const float* ends, int64_t num_rays,
                        int32_t num_samples, float* density, cn_stream_t stream);

/* FruitField.forward (fruit_nerf/fruit_field.py:284-302) at sample mid-points.
 * Outputs per sample: density [R,S], rgb [R,S,3], semantics logit [R,S], positions [R,S,3] (NULL to skip).
 * This is also FruitModel.get_export_outputs' field part (fruit_nerf/fruit_nerf.py:476-494). */
int cn_field_eval(const cn_field_params* params, const cn_scene* scene, int32_t app_mode, int32_t sh_unit_dir,
                  const float* origins, const float* directions, const int64_t* camera_indices,
                  const float* starts, const float* ends, int64_t num_rays, int32_t num_samples,
                  float* density, float* rgb, float* semantics, float* positions, cn_stream_t stream);
/* The same with the matrix arithmetic of the caller's choice (cn_field_eval = CN_MATRIX_FP32), as cn_render_opts.matrix_precision
 * chooses it for the fused renderers: CN_MATRIX_SPLIT_BF16 evaluates the two field shapes of the reference's method configs
 * (fruit_nerf_method and fruit_nerf_method_big / _huge, fruit_nerf/fruit_nerf_config.py:29-172) with every matrix operand as
 * bf16 hi + lo and fp32 accumulation -- the path the _big / _huge models render and export through; CN_MATRIX_F16 takes the same
 * kernel (there is no fp16-operand form of the shape-generic evaluation); any other shape is evaluated in exact fp32. */
int cn_field_eval_mp(const cn_field_params* params, const cn_scene* scene, int32_t app_mode, int32_t sh_unit_dir,
                     const float* origins, const float* directions, const int64_t* camera_indices,
                     const float* starts, const float* ends, int64_t num_rays, int32_t num_samples,
                     float* density, float* rgb, float* semantics, float* positions, int32_t matrix_precision,
                     cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Compositing
 * ------------------------------------------------------------------------------------------- */

/* RaySamples.get_weights + RGB/Accumulation/Depth(median)/Semantic renderers + colormap
 * (fruit_nerf/fruit_nerf.py:556-597).  rgb/semantics inputs may be NULL (then those outputs are skipped:
 * proposal levels only need weights + depth).  Outputs [R,3],[R,1],[R,1],[R,1],[R,3],[R,S]; any NULL skips. */
int cn_composite(const float* starts, const float* ends, const float* density, const float* rgb,
                 const float* semantics, int64_t num_rays, int32_t num_samples, int32_t bg_mode,
                 const float* bg_color_host, int32_t eval_clamp, float* out_rgb, float* out_accumulation,
                 float* out_depth, float* out_semantics, float* out_semantics_colormap, float* out_weights,
                 cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused renderer (the hot path): sampler + FruitField + compositing in one launch
 * ------------------------------------------------------------------------------------------- */

/* Replaces, for one chunk of rays, FruitModel.get_outputs / get_inference_outputs after sampling
 * (fruit_nerf/fruit_nerf.py:551-597, 503-539) and get_density_for_camera_ray_bundle (:337-342).
 * `bins` = euclidean sample bins [R,S+1] from the proposal sampler, or NULL to sample S bins with
 * opts->spacing between nears and fars (UniformSamplerWithNoise in eval, fruit_nerf.py:188).
 * No [R,S,.] tensor is materialised unless out_weights is given.  Outputs as cn_composite. */
size_t cn_render_workspace_bytes(const cn_field_params* params);
int cn_render_rays(const cn_field_params* params, const cn_scene* scene, const cn_render_opts* opts,
                   const float* origins, const float* directions, const float* nears, const float* fars,
                   const int64_t* camera_indices, const float* bins, int64_t num_rays, float* out_rgb,
                   float* out_accumulation, float* out_depth, float* out_semantics,
                   float* out_semantics_colormap, float* out_weights, void* workspace, size_t workspace_bytes,
                   cn_stream_t stream);

/* Same kernel with per-sample outputs instead of compositing: FruitModel.get_export_outputs
 * (fruit_nerf/fruit_nerf.py:476-494). density [R,S], rgb [R,S,3], semantics [R,S], positions [R,S,3],
 * semantics_colormap [R,S] int64 labels = heaviside(sigmoid(sem) - 0.9, 0) (:488-492); any output may be NULL. */
int cn_render_samples(const cn_field_params* params, const cn_scene* scene, const cn_render_opts* opts,
                      const float* origins, const float* directions, const float* nears, const float* fars,
                      const int64_t* camera_indices, const float* bins, int64_t num_rays, float* density,
                      float* rgb, float* semantics, float* positions, int64_t* semantics_colormap, void* workspace,
                      size_t workspace_bytes, cn_stream_t stream);

/* Fused proposal sampler: piecewise initial samples -> proposal net 0 -> PDF -> proposal net 1 -> PDF
 * (ProposalNetworkSampler as configured at fruit_nerf/fruit_nerf.py:157-164, eval mode).
 * s_prop: host [num_levels] samples per proposal level (256, 96); s_final: field samples.
 * Outputs: euclidean bins [R,s_final+1], spacing bins [R,s_final+1] (NULL to skip),
 * prop_depth [num_levels][R] median depths (NULL to skip). */
size_t cn_proposal_sample_workspace_bytes(int64_t num_rays, const int32_t* s_prop_host, int32_t num_levels,
                                          int32_t s_final);
int cn_proposal_sample(const cn_density_params* const* props_host, int32_t num_levels, const cn_scene* scene,
                       const float* origins, const float* directions, const float* nears, const float* fars,
                       int64_t num_rays, const int32_t* s_prop_host, int32_t s_final, float anneal,
                       float* euclidean_bins, float* spacing_bins, float* prop_depth, void* workspace,
                       size_t workspace_bytes, cn_stream_t stream);

/* The same with the arithmetic of the proposal networks chosen like cn_render_opts.matrix_precision: CN_MATRIX_F16 on half
 * tables (CN_TABLE_F16) evaluates them as tiny-cuda-nn does under the method's mixed_precision=True (fruit_nerf_config.py:35;
 * HashMLPDensityField with implementation="tcnn", fruit_nerf.py:133-142) -- packed-fp16 grid interpolation, fp16 weights and
 * layer inputs, fp32 accumulation, fp16 network output; every other combination computes as cn_proposal_sample does. */
int cn_proposal_sample_mp(const cn_density_params* const* props_host, int32_t num_levels, const cn_scene* scene,
                          const float* origins, const float* directions, const float* nears, const float* fars,
                          int64_t num_rays, const int32_t* s_prop_host, int32_t s_final, float anneal,
                          float* euclidean_bins, float* spacing_bins, float* prop_depth, int32_t matrix_precision,
                          cn_stream_t stream);

/* The same sampler under model.train() (fruit_nerf/fruit_nerf.py:549 -> ProposalNetworkSampler.generate_ray_samples with
 * the samplers' train_stratified / single_jitter defaults, components/ray_samplers.py:84-87): level-0 bins jittered by ONE
 * uniform random per ray, every PDF resampling at u + rand / nb with one random per ray, and every level's spacing bins,
 * euclidean intervals and densities written out -- what interlevel_loss (fruit_nerf.py:608-611) and the proposal
 * backward read.  jitter: device [num_levels + 1][R] uniforms in [0, 1) (row l + 1 = the resampling after level l).
 * final_starts / final_ends: euclidean_bins[:, :-1] / [:, 1:] as contiguous arrays (what the field kernels take).
 * Replaces the composed cn_sample_spaced / cn_proposal_density / cn_composite / cn_sample_pdf chain of the training
 * forward (same arithmetic per step; tested against it and against oracle/losses.py). */
typedef struct cn_proposal_level_out {
  float* spacing_bins; /* [R, S_l + 1] */
  float* starts;       /* [R, S_l] euclidean */
  float* ends;         /* [R, S_l] */
  float* density;      /* [R, S_l] */
} cn_proposal_level_out;
int cn_proposal_sample_train(const cn_density_params* const* props_host, int32_t num_levels, const cn_scene* scene,
                             const float* origins, const float* directions, const float* nears, const float* fars,
                             int64_t num_rays, const int32_t* s_prop_host, int32_t s_final, float anneal,
                             const float* jitter, const cn_proposal_level_out* levels_host, float* euclidean_bins,
                             float* spacing_bins, float* final_starts /*[R,s_final] or NULL*/,
                             float* final_ends /*[R,s_final] or NULL*/, cn_stream_t stream);
/* The same with the annealing exponent in device memory (anneal_dev [1]): for a captured HIP graph of the training iteration. */
int cn_proposal_sample_train_dev(const cn_density_params* const* props, int32_t num_levels, const cn_scene* scene,
                                 const float* origins, const float* directions, const float* nears, const float* fars,
                                 int64_t num_rays, const int32_t* s_prop, int32_t s_final, const float* anneal_dev,
                                 const float* jitter, const cn_proposal_level_out* levels, float* euclidean_bins,
                                 float* spacing_bins, float* final_starts, float* final_ends, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Exporters
 * ------------------------------------------------------------------------------------------- */

/* sample_volume masking + compaction (fruit_nerf/export/exporter_utils.py:100-153) over N samples:
 *   set 0 "semantic_colormap": label(sigmoid(sem) > 0.9) and density >= den_thresh, 4th colour sigmoid(sem)
 *   set 1 "semantic":          sem >= sem_thresh and density >= den_thresh,        4th colour sigmoid(sem)
 *   set 2 "density":           density >= den_thresh,                              4th colour sigmoid(density)
 * Appends to points[set] ([cap,3]) / colors[set] ([cap,4]) at counts[set] (device int64[3], caller zeroes);
 * rows beyond `capacity` are counted but not written.  Order within a set is not the reference's
 * (atomic append); compare as sets. */
int cn_export_compact(const float* positions, const float* rgb, const float* semantics, const float* density,
                      int64_t num_samples, float sem_thresh, float den_thresh, int64_t capacity,
                      float* const* points_host3, float* const* colors_host3, int64_t* counts,
                      cn_stream_t stream);

/* generate_point_cloud inner step (fruit_nerf/export/exporter_utils_nerfacto.py:156-166):
 * point = o + d*depth kept where semantics_colormap[:,0] > 0; appends xyz/rgb/view-dir rows. */
int cn_pointcloud_compact(const float* origins, const float* directions, const float* depth, const float* rgb,
                          const float* semantics_colormap, int64_t num_rays, int64_t capacity, float* points,
                          float* colors, float* view_dirs, int64_t* count, cn_stream_t stream);

/* generate_point_cloud with SEVERAL of the reference's calls per launch (exporter_utils_nerfacto.py:125-183: a call =
 * rays_per_call random pixels -> render -> append -> `while not finished` check; 2 048 rays at debug/exporter_nerfacto.py:91,
 * 32 768 upstream).  The cloud the loop produces is the union of the kept points of calls 0 .. c*, c* = the first call at
 * which the running count reaches the target -- however the calls are grouped into launches, provided that
 *   cn_pixel_sample            draws call c's pixels from a counter-based stream: ray_indices[(c - first) * rays_per_call + r]
 *                              = floor(u * (num_cameras, H, W)), u = hash(seed, c, r, component) -- nerfstudio's PixelSampler
 *                              formula on a stream that does not depend on the grouping (first_call: DEVICE int64, so a HIP
 *                              graph can advance it);
 *   cn_pointcloud_compact_calls  counts the kept rays per call (call_counts [ceil(num_rays / rays_per_call)], written), finds
 *                              on the device the call at which *count reaches target_points, writes the number of rays up
 *                              to the end of that call to *ray_limit (0 when *count had reached the target before) and
 *                              appends exactly those rays' points as cn_pointcloud_compact does. */
int cn_pixel_sample(uint64_t seed, const int64_t* first_call, int32_t num_calls, int32_t rays_per_call, int32_t num_cameras,
                    int32_t height, int32_t width, int64_t* ray_indices /*[num_calls*rays_per_call,3]*/, cn_stream_t stream);
int cn_pointcloud_compact_calls(const float* origins, const float* directions, const float* depth, const float* rgb,
                                const float* semantics_colormap, int64_t num_rays, int64_t rays_per_call,
                                int64_t target_points, int64_t capacity, float* points, float* colors, float* view_dirs,
                                int64_t* count, int64_t* call_counts, int64_t* ray_limit, cn_stream_t stream);

/* Embedding.mean(dim=0) (fruit_nerf/fruit_field.py:220,257): [num_images, dim] -> [dim]. */
int cn_embedding_mean(const float* embedding, int32_t num_images, int32_t dim, float* mean, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training: losses, backward, optimiser (FruitModel.get_loss_dict fruit_nerf/fruit_nerf.py:601-615,
 * optimisers fruit_nerf/fruit_nerf_config.py:45-60).  Gradient buffers use the SAME structs as the
 * parameters (cn_field_params / cn_density_params whose pointers address the gradient arrays) and are
 * ACCUMULATED into (atomics); the caller zeroes them (cn_adam_step can do it).
 * ------------------------------------------------------------------------------------------- */

/* Forward + backward of the volume renderer and of the rgb (MSE over R*3) and semantics
 * (semantic_loss_weight * BCE-with-logits mean over R) losses, training mode (no clamp, "last_sample"
 * background, semantics rendered with detached weights: fruit_nerf.py:556-591).  Inputs per sample [R,S]:
 * starts, ends, density, rgb [R,S,3], semantics.  Outputs: rendered rgb [R,3] / semantics [R,1] /
 * accumulation [R,1] / weights [R,S] (any NULL skips), gradients d_density [R,S], d_rgb [R,S,3],
 * d_semantics [R,S] of (rgb_loss + semantics_loss), and loss_sums[0] += sum (rgb-image)^2,
 * loss_sums[1] += sum BCE terms (divide by 3R / R on the host).
 * spacing_bins (may be NULL) [R,S+1]: the sampler's bins in the spacing domain; when given, loss_sums has FIVE slots and
 * loss_sums[4] += the sum over rays of nerfstudio's distortion_loss of these weights (the "distortion" entry of
 * get_metrics_dict, fruit_nerf.py:643; what cn_distortion_metric computes pair by pair, here from two more prefix sums). */
int cn_train_render_backward(const float* starts, const float* ends, const float* density, const float* rgb,
                             const float* semantics, const float* image /*[R,3]*/,
                             const float* fruit_mask /*[R,1]*/, int64_t num_rays, int32_t num_samples,
                             float semantic_loss_weight, float* out_rgb, float* out_semantics,
                             float* out_accumulation, float* out_weights, float* d_density, float* d_rgb,
                             float* d_semantics, float* loss_sums, const float* spacing_bins, cn_stream_t stream);

/* nerfstudio interlevel_loss term of ONE proposal level (fruit_nerf.py:609-612): final spacing bins
 * [R,Sf+1] and (detached) final weights [R,Sf] against the level's spacing bins [R,Sp+1] and density
 * [R,Sp] (with its euclidean starts/ends).  Writes d loss / d proposal density [R,Sp] for
 * loss = loss_mult * mean_{R,Sf}(...), and adds the un-normalised sum to *loss_sum. */
int cn_interlevel_backward(const float* final_spacing_bins, const float* final_weights,
                           const float* prop_spacing_bins, const float* prop_starts, const float* prop_ends,
                           const float* prop_density, int64_t num_rays, int32_t s_final, int32_t s_prop,
                           float loss_mult, float* d_prop_density, float* loss_sum, cn_stream_t stream);
/* The same for every proposal level of an iteration (1..4) in ONE launch: the levels share the final bins / weights and
 * *loss_sum (the reference sums the levels' terms into one "interlevel_loss", fruit_nerf.py:609-612). */
typedef struct cn_interlevel_level {
  const float* spacing_bins; /* [R,Sp+1] */
  const float* starts;       /* [R,Sp] euclidean */
  const float* ends;
  const float* density;      /* [R,Sp] */
  float* d_density;          /* [R,Sp] out */
  int32_t num_samples;       /* Sp */
  int32_t reserved;
} cn_interlevel_level;
int cn_interlevel_backward_levels(const float* final_spacing_bins, const float* final_weights,
                                  const cn_interlevel_level* levels, int32_t num_levels, int64_t num_rays, int32_t s_final,
                                  float loss_mult, float* loss_sum, cn_stream_t stream);

/* Parameter gradients of FruitField (training branch: per-camera appearance, semantic MLP on detached geo
 * features, fruit_nerf/fruit_field.py:235-282) from per-sample upstream gradients; the forward is recomputed
 * tile by tile.  app_mean [app_dim] is read with CN_APP_MEAN only.  d_positions / d_directions (optional) receive
 * d loss / d (world sample position) and d loss / d (ray direction, through the SH colour input) per sample -- the
 * inputs of the camera pose refinement's backward (the positions get `requires_grad`, fruit_field.py:181-184). */
int cn_field_backward(const cn_field_params* params, const cn_field_params* grads, const cn_scene* scene,
                      int32_t app_mode, int32_t sh_unit_dir, const float* app_mean, const float* origins,
                      const float* directions, const int64_t* camera_indices, const float* starts,
                      const float* ends, const float* d_density, const float* d_rgb, const float* d_semantics,
                      int64_t num_rays, int32_t num_samples, float* d_positions /*[R,S,3] or NULL*/,
                      float* d_directions /*[R,S,3] or NULL*/, cn_stream_t stream);
/* The same with the matrix arithmetic of the caller's choice (cn_field_backward = CN_MATRIX_FP32).  CN_MATRIX_F16: the
 * reference's training arithmetic class -- it trains under mixed_precision=True on tiny-cuda-nn's fp16 modules
 * (fruit_nerf/fruit_nerf_config.py:35, fruit_field.py:95,125-167): the forward recompute with fp16 weights and layer inputs
 * (what cn_render_samples computes in that mode), the gradient products dX and dW with bf16 operands (fp32's exponent range:
 * no loss scale), fp32 accumulation throughout, fp32 master gradients out.  CN_MATRIX_SPLIT_BF16 (a ~fp32 forward): the
 * exact-fp32 kernel. */
int cn_field_backward_mp(const cn_field_params* params, const cn_field_params* grads, const cn_scene* scene,
                         int32_t app_mode, int32_t sh_unit_dir, const float* app_mean, const float* origins,
                         const float* directions, const int64_t* camera_indices, const float* starts,
                         const float* ends, const float* d_density, const float* d_rgb, const float* d_semantics,
                         int64_t num_rays, int32_t num_samples, float* d_positions, float* d_directions,
                         int32_t matrix_precision, cn_stream_t stream);

/* The same for the other field shapes of the reference's method configs (fruit_nerf_method_big / _huge:
 * fruit_nerf/fruit_nerf_config.py:66-172): base MLP 2 layers, semantic MLP 2-3 layers, colour MLP 3 layers, widths
 * <= 128.  Needs a zero-initialised-by-the-callee workspace of cn_field_backward_general_workspace_bytes(params)
 * (per-workgroup partial weight gradients); d_positions / d_directions as for cn_field_backward. */
size_t cn_field_backward_general_workspace_bytes(const cn_field_params* params);
int cn_field_backward_general(const cn_field_params* params, const cn_field_params* grads, const cn_scene* scene,
                              int32_t app_mode, int32_t sh_unit_dir, const float* app_mean, const float* origins,
                              const float* directions, const int64_t* camera_indices, const float* starts,
                              const float* ends, const float* d_density, const float* d_rgb,
                              const float* d_semantics, int64_t num_rays, int32_t num_samples,
                              float* d_positions /*[R,S,3] or NULL*/, float* d_directions /*[R,S,3] or NULL*/,
                              void* workspace, size_t workspace_bytes, cn_stream_t stream);

/* Parameter gradients of one proposal network from d loss / d density [R,S]; d_positions as above. */
int cn_proposal_backward(const cn_density_params* params, const cn_density_params* grads, const cn_scene* scene,
                         const float* origins, const float* directions, const float* starts, const float* ends,
                         const float* d_density, int64_t num_rays, int32_t num_samples,
                         float* d_positions /*[R,S,3] or NULL*/, cn_stream_t stream);

/* Camera pose refinement backward (camera_optimizer.apply_to_raybundle, fruit_nerf/fruit_nerf.py:547, trained through
 * the "camera_opt" group, fruit_nerf.py:195).  Step 1, per sample set: ACCUMULATE per-ray
 * d_origins += sum_s d_positions, d_directions += sum_s mid_s d_positions (+ sum_s d_dir_samples when not NULL). */
int cn_ray_backward(const float* d_positions /*[R,S,3]*/, const float* d_dir_samples /*[R,S,3] or NULL*/,
                    const float* starts, const float* ends, int64_t num_rays, int32_t num_samples,
                    float* d_origins /*[R,3]*/, float* d_directions /*[R,3]*/, cn_stream_t stream);

/* Step 2: chain through exp_map_SO3xR3 (o' = o + t, d' = R(w) d): ACCUMULATE into grad_pose [C,6].
 * directions_raw = the ray directions BEFORE cn_apply_pose_adjustment.  num_cameras = C (every camera index must lie in
 * [0, C)): up to 2048 cameras the rays of a workgroup are summed per camera on chip before they reach grad_pose; 0 = not
 * given (one atomic per ray and entry). */
int cn_pose_adjustment_backward(const float* pose_adjustment /*[C,6]*/, const int64_t* camera_indices,
                                const float* directions_raw /*[R,3]*/, const float* d_origins,
                                const float* d_directions, int64_t num_rays, int32_t num_cameras, float* grad_pose,
                                cn_stream_t stream);

/* get_loss_dict (fruit_nerf/fruit_nerf.py:601-615) + the scalar metrics of get_metrics_dict (:639-645) from the loss sums
 * the training kernels left in loss_sums = {sum (rgb - image)^2, sum BCE, sum interlevel terms, camera regulariser} and, when
 * num_sums = 5, the sum cn_distortion_metric accumulated:
 * out[8] = {rgb_loss, semantics_loss, interlevel_loss, camera_opt_regularizer, psnr, |translations|, |rotations|, distortion}
 * (pose_adjustment NULL: the two norms are 0; distortion = loss_sums[4] / R, the "distortion" entry of get_metrics_dict, or 0
 * when num_sums = 4).  One launch instead of a dozen one-element host-composed kernels. */
int cn_train_epilogue(const float* loss_sums, int32_t num_sums /*4 or 5*/, int64_t num_rays, int32_t num_samples,
                      float semantic_loss_weight, float interlevel_loss_mult, const float* pose_adjustment /*[C,6] or NULL*/,
                      int32_t num_cameras, float* out, cn_stream_t stream);

/* camera_opt_regularizer of CameraOptimizer.get_loss_dict (fruit_nerf/fruit_nerf.py:614):
 * mean_c |t_c| * trans_l2_penalty + mean_c |w_c| * rot_l2_penalty; adds the loss to *loss_out and, when grad_pose
 * is not NULL, its gradient to grad_pose. */
int cn_pose_regularizer(const float* pose_adjustment, int32_t num_cameras, float trans_l2_penalty,
                        float rot_l2_penalty, float* grad_pose, float* loss_out, cn_stream_t stream);

/* nerfstudio distortion_loss of the final level (the "distortion" entry of get_metrics_dict,
 * fruit_nerf/fruit_nerf.py:643): adds sum over rays to *sum_out (divide by R on the host). */
int cn_distortion_metric(const float* spacing_bins /*[R,S+1]*/, const float* weights /*[R,S]*/, int64_t num_rays,
                         int32_t num_samples, float* sum_out, cn_stream_t stream);

/* torch.optim.Adam step (no weight decay / amsgrad) on one flat tensor; `step` is 1-based;
 * zero_grad != 0 clears the gradient after use. */
int cn_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step, double lr,
                 double beta1, double beta2, double eps, int32_t zero_grad, cn_stream_t stream);

/* torch.optim.RAdam step (fruit_nerf_method_big / _huge: fruit_nerf/fruit_nerf_config.py:101-117,151-167; no weight
 * decay), same arguments as cn_adam_step. */
int cn_radam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step, double lr,
                 double beta1, double beta2, double eps, int32_t zero_grad, cn_stream_t stream);

/* The per-step scalars of cn_adam_step as a host computation (hyper_host [8] = {lr / bias correction 1, beta1, beta2, 1 - beta1,
 * 1 - beta2, 1 / sqrt(bias correction 2), eps, 0}) and the update reading them from DEVICE memory: a HIP graph that captured a
 * whole training iteration replays with the schedules' new values after one small host-to-device copy. */
int cn_adam_hyper(int32_t step, double lr, double beta1, double beta2, double eps, float* hyper_host);
int cn_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper /*device [8]*/,
                     int32_t zero_grad, cn_stream_t stream);
/* Several optimiser groups of one flat buffer in ONE launch: group k owns elements [bounds_host[k], bounds_host[k + 1]) (a HOST
 * array of num_groups + 1 ascending offsets from 0, at most 8 groups), hyper [num_groups, 8] on the device as above, with
 * hyper[k][7] != 0 marking a group that takes no step this iteration.  Gradients are zeroed everywhere. */
int cn_adam_step_groups_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* bounds_host,
                            int32_t num_groups, const float* hyper /*device [num_groups, 8]*/, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Depth-based semantic projection (the alternative to the NeRF projection:
 * fruit_nerf/scripts/depth_based_semantic_projection.py).  float64 geometry, float32 z-buffer, uint8 label
 * image, all [height, width] row-major -- the reference's 1440 x 1920 arrays.
 * ------------------------------------------------------------------------------------------- */

/* get_projection (:45-49) + the pixel arithmetic of update_buffer (:85-89): im = P @ [p, 1];
 * yx = round(im[:2] / -im[2]) half-to-even; ys = clip(yx[0], 0, width-1) (column), xs = clip(yx[1], 0, height-1)
 * (row); zs = -im[2].  P [3,4] and points [N,3] are float64 device arrays. */
int cn_depth_project(const double* P, const double* points, int64_t num_points, int32_t height, int32_t width,
                     int32_t* xs, int32_t* ys, double* zs, cn_stream_t stream);

size_t cn_zbuffer_workspace_bytes(int32_t height, int32_t width);

/* update_buffer(large=True) (:90-94): img[xs, ys] = label; z_buffer[xs, ys] = zs -- the LAST point of a pixel wins. */
int cn_zbuffer_update_large(const int32_t* xs, const int32_t* ys, const double* zs, int64_t num_points, int32_t label,
                            int32_t height, int32_t width, float* z_buffer, uint8_t* img, void* workspace,
                            size_t workspace_bytes, cn_stream_t stream);

/* update_buffer(large=False) (:95-105): a point is accepted when z <= z_buffer at its pixel; accepted pixels take the
 * point's z and the label.  visible (optional) = 255 on the accepted pixels, 0 elsewhere (what the caller paints into
 * occ_free_*.png, :159-161). */
int cn_zbuffer_update(const int32_t* xs, const int32_t* ys, const double* zs, int64_t num_points, int32_t label,
                      int32_t height, int32_t width, float* z_buffer, uint8_t* img, uint8_t* visible, void* workspace,
                      size_t workspace_bytes, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Statistical outlier removal of the point-cloud exporter: the k-nearest-neighbour pass of open3d's
 * remove_statistical_outlier(nb_neighbors=20, std_ratio) as called at
 * fruit_nerf/export/exporter_utils_nerfacto.py:194-199.  The caller bins the points on a uniform grid
 * (points sorted by linear cell index (z*gy + y)*gx + x; cell_start [gx*gy*gz + 1] offsets); mean_distance[i]
 * = mean of the distances from sorted point i to its nb_neighbors nearest points, itself (0) included.
 * ------------------------------------------------------------------------------------------- */
int cn_knn_mean_distance(const float* points_sorted, const int32_t* cell_start, int32_t gx, int32_t gy, int32_t gz,
                         float origin_x, float origin_y, float origin_z, float cell_size, int64_t num_points,
                         int32_t nb_neighbors, float* mean_distance, cn_stream_t stream);

/* The same two passes on a TWO-LEVEL grid (round 5).  An exported cloud is surfaces -- or clumps: the cells of any dense grid
 * that fits memory then hold thousands of points each.  cn_point_grid: a dense TOP grid (top[0] x top[1] x top[2] cells of
 * top_cell_size, at most 1024 per axis) whose occupied cells are numbered in top_rank [top[2]*top[1]*top[0]] (linear index
 * (z*top[1] + y)*top[0] + x; -1 = empty), and sub^3 FINE cells of top_cell_size / sub inside every occupied top cell: the points
 * are sorted by (rank of their top cell) * sub^3 + ((z % sub)*sub + y % sub)*sub + x % sub of their fine cell, and cell_start
 * [occupied * sub^3 + 1] holds the offsets.  A query searches fine rings (at most fine_rings of them), then -- an isolated
 * point -- whole top cells.  Same results as the dense-grid entry points (the exact k nearest). */
typedef struct cn_point_grid {
  int32_t top[3];
  int32_t sub;
  int32_t fine_rings;
  float origin[3];
  float top_cell_size;
  const int32_t* top_rank;
  const int32_t* cell_start;
} cn_point_grid;
int cn_knn_mean_distance_grid(const float* points_sorted, const cn_point_grid* grid, int64_t num_points, int32_t nb_neighbors,
                              float* mean_distance, cn_stream_t stream);
int cn_estimate_normals_grid(const float* points_sorted, const cn_point_grid* grid, int64_t num_points, int32_t knn,
                             double* normals, int32_t* degenerate, cn_stream_t stream);

/* Normals of the exported cloud: open3d's PointCloud::EstimateNormals() at its defaults, as generate_point_cloud calls it
 * for `ns-export pointcloud --normal-method open3d` (fruit_nerf/export/exporter_utils_nerfacto.py:203-212; README.md:125).
 * Same grid as cn_knn_mean_distance.  normals [N,3] DOUBLE (sorted order): unit eigenvector of the smallest eigenvalue of
 * the covariance of the point's `knn` nearest points (itself included; open3d: 30), sign as the solver leaves it -- the
 * caller re-orients against the view directions (:219-225).  degenerate [N] (optional): 1 where fewer than three neighbours
 * exist or the covariance is zero -- the normal is (0, 0, 1) there, as open3d's. */
int cn_estimate_normals(const float* points_sorted, const int32_t* cell_start, int32_t gx, int32_t gy, int32_t gz,
                        float origin_x, float origin_y, float origin_z, float cell_size, int64_t num_points, int32_t knn,
                        double* normals, int32_t* degenerate, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Super-cluster stage of the segmenter (segmentation/segmenter.py:69-86, get_super_clusters):
 * voxel_down_sample -> cluster_dbscan(eps, min_points) -> remove_statistical_outlier (cn_knn_mean_distance).
 * The caller bins (sorts) the points; see each entry point.
 * ------------------------------------------------------------------------------------------- */

/* open3d voxel_down_sample: out[s] = mean of the rows [segment_start[s], segment_start[s+1]) of
 * values_sorted [N, channels] (rows sorted by voxel). */
int cn_segment_mean(const float* values_sorted, const int32_t* segment_start, int64_t num_segments, int32_t channels,
                    float* out, cn_stream_t stream);

/* DBSCAN (open3d cluster_dbscan(eps, min_points) as called at segmentation/segmenter.py:76-77) on a sparse grid whose
 * cells have a diagonal <= eps (cell_size * sqrt(3) <= eps; grid dimensions < 2^20 per axis).  The caller sorts the
 * points by cell key (z*dim_y + y)*dim_x + x and passes the OCCUPIED cells: cell_keys [num_cells] ascending,
 * cell_start [num_cells + 1] offsets into the sorted points, point_cell [N] = occupied-cell index of each sorted point.
 * order [N] = original index of each sorted point (ties between clusters for a border point go to the core neighbour
 * with the smallest original index).  neighbour_count [N] (number of points within eps, itself included, saturated at
 * min_points: == min_points marks the core points) and parent [N] are work arrays; root [N] = sorted index of the
 * representative of the point's cluster, -1 for noise. */
size_t cn_dbscan_workspace_bytes(int64_t num_cells);
int cn_dbscan(const float* points_sorted, const int64_t* cell_keys, const int32_t* cell_start, const int32_t* point_cell,
              int64_t num_cells, int64_t dim_x, int64_t dim_y, int64_t dim_z, float cell_size, float eps,
              int32_t min_points, const int64_t* order, int64_t num_points, int32_t* neighbour_count, int32_t* parent,
              int32_t* root, void* workspace, size_t workspace_bytes, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Image stage of the merger (segmentation/merger.py:219-271) on a stack of gray images [num_images, height, width]
 * (the projections of a15, quantised as save_image does).  Per image, inside roi (x0, y0, x1, y1; NULL = whole image):
 * foreground = gray > thresh; the contour of largest area as cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) +
 * max(key=cv2.contourArea) select it (always an outer border; ties to the component found last in raster order):
 *   area [J]        cv2.contourArea of it (0 when the image has no foreground)          -- get_wo_occlusion_projection_area
 *   bbox [J,4]      cv2.boundingRect: x, y, w, h in full-image coordinates
 *   start [J]       linear index of the contour's first pixel, -1 when none (optional)
 * and, when `labels` ([*, height, width] uint8 instance-label frames, image j uses labels[label_index[j]]) is given
 * -- get_visible_projection_area, whose drawContours(mask, cnt, -1, 255, -1) call with a BARE contour draws its vertices only:
 *   vertex_count [J] number of distinct vertex pixels of the compressed contour ("area" of the visible projection)
 *   label [J], label_count [J]  the label under most of those pixels (ties to the larger label) and that count.
 * ------------------------------------------------------------------------------------------- */
size_t cn_contour_workspace_bytes(int32_t num_images, int32_t height, int32_t width);
int cn_contour_largest(const uint8_t* gray, const int32_t* roi, int32_t num_images, int32_t height, int32_t width,
                       int32_t thresh, const uint8_t* labels, const int32_t* label_index, float* area, int32_t* bbox,
                       int32_t* start, int32_t* vertex_count, int32_t* label, int32_t* label_count, void* workspace,
                       size_t workspace_bytes, cn_stream_t stream);

/* K-means sub-clustering of one super-cluster (segmentation/segmenter.py:28-45: sklearn KMeans(init="k-means++",
 * n_clusters=k, n_init="auto", random_state=0), called per super-cluster at :153-181).  ONE Lloyd iteration in float64, as
 * sklearn's _kmeans_single_lloyd: labels[i] <- index of the nearest of the k centres [k,3] (first minimum); *changed |= 1
 * when any label differs from its previous value; with accumulate != 0 the coordinate sums [k,3] and counts [k] of the new
 * assignment are ADDED to `sums` / `counts` (caller zeroes them).  The k-means++ seeding and the stopping rules stay on
 * the host. points [N,3] float64 (centred as sklearn does), k <= 32. */
int cn_kmeans_step(const double* points, int64_t num_points, const double* centers, int32_t k, int32_t* labels,
                   double* sums, int64_t* counts, int32_t* changed, int32_t accumulate, cn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Deterministic gradient accumulation -- a TEST mode, a second library built from the same sources
 * (libcropnerf_hip_det.so: build.py compiles the training units once more with -DCN_DETERMINISTIC_SCATTER=1).
 * The reference has no counterpart (torch's float atomics are as order-dependent as these); the mode exists so that the
 * tests of the training step (graph replay against eager launches, a resumed run against a straight one:
 * fruit_nerf/fruit_nerf.py:543-615 through the trainer) can compare parameters and Adam moments BIT FOR BIT instead of
 * within the per-cent noise of float-atomic summation orders.  In that library every global float atomic of the training
 * kernels adds round(v * 2^44) to a 64-bit integer shadow of its destination, and a flush pass behind every such kernel adds
 * the shadows to the floats in a fixed order (csrc/cn_det.hpp).  Not the default build's arithmetic (values below 5.7e-14
 * vanish, |sum| < 5.2e5) and several times slower: never selected by the product path.
 *   cn_deterministic_build()     1 in the test library, 0 in the default one.
 *   cn_deterministic_register()  host call, not during a stream capture: the `count` floats at `base` (device) accumulate
 *                                through `shadow` (device, `count` zeroed int64, caller-owned like every buffer); at most 16
 *                                disjoint ranges.  `miss_counter` (device uint64, may be NULL; the last one given is
 *                                used) counts atomics whose destination lies in no registered range -- those fall back to the
 *                                float atomic.  Default build: CN_ERR_UNSUPPORTED.
 *   cn_deterministic_clear()     forget every range (buffers were re-allocated).
 *   cn_deterministic_flush()     the pass the library enqueues by itself behind each accumulating kernel; exported for
 *                                callers that write shadows of their own.
 * ------------------------------------------------------------------------------------------- */
int cn_deterministic_build(void);
int cn_deterministic_register(float* base, int64_t count, int64_t* shadow, uint64_t* miss_counter);
int cn_deterministic_clear(void);
int cn_deterministic_flush(cn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CROPNERF_HIP_H */

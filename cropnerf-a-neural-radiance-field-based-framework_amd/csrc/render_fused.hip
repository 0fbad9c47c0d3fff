// render_fused.hip -- the hot path: sampler + FruitField + alpha compositing in ONE launch, nothing [R,S,.]-shaped
// ever touches HBM.  Replaces fruit_nerf/fruit_nerf.py:551-597 (get_outputs after sampling), :503-539
// (get_inference_outputs), :337-342 (get_density_for_camera_ray_bundle) and :476-494 (get_export_outputs).
//
// Mapping (gfx950, wave64):
//   * one wavefront owns one ray at a time and walks its samples in chunks of 64;
//   * hash-grid phase: lane (g = lane>>4, j = lane&15) gathers levels 4g..4g+3 for the four samples 16c+j (c=0..3) of
//     the chunk, so a lane's 8 features per sample are exactly the B-operand rows it must feed to the first MFMA
//     (k = g per step) -- no transpose, no LDS round trip for activations;
//   * MLPs: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).  Weights are the A operand (rows = output neurons),
//     the 16 samples of a column tile are the B operand / the accumulator columns, so a layer's accumulator
//     registers ARE the next layer's B operands: step (t,r) of the next layer consumes register r of row tile t,
//     and the weight image in LDS is pre-permuted to that k order (prep kernel below);
//   * algebra done once per launch in the prep kernel: the last Linear of mlp_semantics is folded into the
//     Linear(64,1) head (no activation between them, fruit_field.py:146-157); per ray, the SH and appearance
//     columns of mlp_head's first layer collapse into a 64-vector bias (they are constant along a ray);
//   * compositing: lane l takes sample l of the chunk, wave-level scans give transmittance and the median depth.
//
// Supported field shape: the default `fruit_nerf_method` (fruit_nerf_config.py:29-65): 16 levels x 2 features,
// 32->64->16, 15->64->64->1, 63->64->64->3, appearance 32.  Other shapes return CN_ERR_UNSUPPORTED (the caller
// then uses cn_field_eval + cn_composite, which are shape-generic).
#include "composite_dev.hpp"

namespace cn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- weight image ("blob") layout, in floats -------------------------------------------------------------------
constexpr int OFF_A0 = 0;       // base L0   [mt 4][sq 2][lane 64][4]   W0[16mt+j][8g + 4sq + e]
constexpr int OFF_A1 = 2048;    // base L1   [sq 4][lane 64][4]         W1[j][16sq + 4g + e]
constexpr int OFF_AS0 = 3072;   // sem L0    [mt 4][lane 64][4]         Ws0[16mt+j][4g+e-1] (0 for base neuron 0)
constexpr int OFF_AC0 = 4096;   // colour L0 [mt 4][lane 64][4]         Wc0[16mt+j][16 + 4g+e-1]
constexpr int OFF_AC1 = 5120;   // colour L1 [mt 4][sq 4][lane 64][4]   Wc1[16mt+j][16sq + 4g + e]
constexpr int OFF_B0 = 9216;    // [64]
constexpr int OFF_B1 = 9280;    // [16]
constexpr int OFF_BS0 = 9296;   // [64]
constexpr int OFF_BC1 = 9360;   // [64]
constexpr int OFF_WF = 9424;    // [64] folded semantic head  Wh . Ws1
constexpr int OFF_WRGB = 9488;  // [3][64]
constexpr int OFF_MISC = 9680;  // [0] folded semantic bias, [1..3] rgb bias
constexpr int OFF_WSH = 9688;   // [64][16] Wc0[:, 0:16]
constexpr int OFF_SCALE = 10712;  // [16] hash-grid level scalings (lane group g reads [4g, 4g+4))
constexpr int OFF_LVL = 10728;    // [16][4] unsigned: per-level {first entry, index mask, y multiplier, z multiplier} (cn::Lvl)
constexpr int BLOB_FLOATS = 10728 + 64;
static_assert(BLOB_FLOATS % 4 == 0, "blob is copied as float4");
static_assert(OFF_A0 == 0 && OFF_B0 == 18 * 512, "the split-bf16 images reuse the fp32 A-operand region block for block");
constexpr int WAVE_SCRATCH = 200;  // floats of per-wave LDS scratch: colour bias [64] | 2 x 66 chunk bin edges
// timing-only ablations (outputs are wrong): 1 = skip the hash-grid gathers / skip the MLPs
#ifndef CN_ABLATE_GATHER
#define CN_ABLATE_GATHER 0
#endif
#ifndef CN_ABLATE_MLP
#define CN_ABLATE_MLP 0
#endif

#ifndef CN_FUSED_WAVES
#define CN_FUSED_WAVES 4
#endif
#ifndef CN_FUSED_MIN_WAVES_PER_SIMD
#define CN_FUSED_MIN_WAVES_PER_SIMD 3
#endif
// 4 waves x 3 blocks/CU measured faster than 8 waves x 2 blocks/CU (3.38 vs 3.06 Gsamples/s at C2)
// 1: the composited full render runs as render_split_kernel (gather waves feeding matrix waves, render_split.hpp);
// CN_FUSED_SPLIT=0 in the environment selects render_fused_kernel<false,false> instead (A/B runs, cross-check tests)
#ifndef CN_FUSED_SPLIT_DEFAULT
#define CN_FUSED_SPLIT_DEFAULT 1
#endif
constexpr int FUSED_WAVES = CN_FUSED_WAVES;  // waves per workgroup (they share one LDS weight image)
constexpr int FUSED_THREADS = FUSED_WAVES * 64;

struct PrepArgs {
  const float *w0, *b0, *w1, *b1;        // base
  const float *ws0, *bs0, *ws1, *bs1;    // semantics
  const float *wh, *bh;                  // semantic head
  const float *wc0, *bc0, *wc1, *bc1, *wc2, *bc2;  // colour
  const float* emb;
  const float* emb_mean;  // [32] (workspace), valid when app_mode == CN_APP_MEAN
  int num_images;
  int app_mode;
  int app_rows;
  GridDev grid;  // per-level records of the blob
  int mm;       // matrix mode of render_split_kernel: MM_FP32 fp32 A operands; MM_BF16 split-bf16 images for
                // v_mfma_f32_16x16x32_bf16; MM_F16 fp16 images for v_mfma_f32_16x16x32_f16 (every weight rounded to fp16)
  float* ext;   // [BF16_EXT_FLOATS] the four split-bf16 image blocks that do not fit the fp32 region
};
constexpr int MM_FP32 = CN_MATRIX_FP32, MM_BF16 = CN_MATRIX_SPLIT_BF16, MM_F16 = CN_MATRIX_F16;

// Split-bf16 A operands (cn_render_opts.matrix_precision = 1).  Block b = one (row tile, K block of 32): [hi | lo][lane 64][8 bf16];
// lane (g, j) holds row 16 mt + j and the eight k values its B operand lane supplies.  B operands are the features of a
// lane (k = 8 g + jj) or two accumulator tiles of the previous layer (jj = 4 (t & 1) + e  <->  k = 16 t + 4 g + e), so no
// activation ever changes lane.  Blocks: 0-3 base L0, 4-5 base L1, 6-9 semantics L0, 10-13 colour L0 (K = 16 real + 16
// zero), 14-21 colour L1.  18 blocks fill the fp32 A region exactly, 4 go to the extension.
constexpr int BF16_BLOCKS = 22, BF16_BLOCK_FLOATS = 512, BF16_EXT_FLOATS = 4 * BF16_BLOCK_FLOATS;

__device__ __forceinline__ float bf16_logical_weight(const PrepArgs& p, int b, int g, int j, int jj) {
  if (b < 4) return p.w0[(16 * b + j) * 32 + 8 * g + jj];
  if (b < 6) return p.w1[j * 64 + 16 * (2 * (b - 4) + (jj >> 2)) + 4 * g + (jj & 3)];
  if (b < 14) {
    if (jj >= 4) return 0.f;
    const int m = 4 * g + jj;  // base output neuron; neuron 0 is the density logit, not a geo feature
    if (m == 0) return 0.f;
    return b < 10 ? p.ws0[(16 * (b - 6) + j) * 15 + (m - 1)] : p.wc0[(16 * (b - 10) + j) * 63 + 16 + (m - 1)];
  }
  const int mt = (b - 14) >> 1, kb = (b - 14) & 1;
  return p.wc1[(16 * mt + j) * 64 + 16 * (2 * kb + (jj >> 2)) + 4 * g + (jj & 3)];
}

__device__ __forceinline__ float bf16_image_word(const PrepArgs& p, int q) {  // q: float index inside the 22-block image
  const int b = q >> 9, half = (q >> 8) & 1, lane = (q >> 2) & 63, w = q & 3;
  unsigned bits[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float x = bf16_logical_weight(p, b, lane >> 4, lane & 15, 2 * w + h);
    const __bf16 hi = (__bf16)x;
    const __bf16 v = half ? (__bf16)(x - (float)hi) : hi;
    bits[h] = (unsigned)__builtin_bit_cast(unsigned short, v);
  }
  return __builtin_bit_cast(float, bits[0] | (bits[1] << 16));
}

// fp16 A operands (cn_render_opts.matrix_precision = CN_MATRIX_F16): the same 22 (row tile, K block of 32) blocks, block b =
// [lane 64][8 halves] = 256 floats at OFF_A0 + 256 b (all of them fit the fp32 region).
__device__ __forceinline__ float f16_image_word(const PrepArgs& p, int q) {
  const int b = q >> 8, lane = (q >> 2) & 63, w = q & 3;
  unsigned bits[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const _Float16 v = (_Float16)bf16_logical_weight(p, b, lane >> 4, lane & 15, 2 * w + h);
    bits[h] = (unsigned)__builtin_bit_cast(unsigned short, v);
  }
  return __builtin_bit_cast(float, bits[0] | (bits[1] << 16));
}
// a value as the fp16 mode sees it (tcnn casts parameters and inputs to half)
__device__ __forceinline__ float f16_round(float x) { return (float)(_Float16)x; }

// Embedding.mean(dim=0) for the 32-wide appearance embedding: 8 partial sums per column, combined through LDS
__global__ void __launch_bounds__(256) prep_mean_kernel(const float* __restrict__ emb, int n, float* __restrict__ mean) {
  __shared__ float part[8][32];
  const int k = threadIdx.x & 31, p = threadIdx.x >> 5;
  float s = 0.f;
  for (int i = p; i < n; i += 8) s += emb[i * 32 + k];
  part[p][k] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += part[q][k];
    mean[k] = t / (float)n;
  }
}

__global__ void __launch_bounds__(256) prep_kernel(PrepArgs p, float* __restrict__ blob, float* __restrict__ app_bias) {
  const int total = BLOB_FLOATS + p.app_rows * 64;
  if (p.mm == MM_BF16)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < BF16_EXT_FLOATS; i += gridDim.x * blockDim.x)
      p.ext[i] = bf16_image_word(p, OFF_B0 + i);
  const bool f16 = p.mm == MM_F16;
  auto R16 = [f16](float x) { return f16 ? f16_round(x) : x; };
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (p.mm == MM_BF16 && i < OFF_B0) {
      v = bf16_image_word(p, i);
    } else if (f16 && i < OFF_B0) {
      v = i < BF16_BLOCKS * 256 ? f16_image_word(p, i) : 0.f;
    } else if (i < OFF_A1) {
      int q = i - OFF_A0;
      int mt = q >> 9, sq = (q >> 8) & 1, lane = (q >> 2) & 63, e = q & 3;
      int g = lane >> 4, j = lane & 15;
      v = p.w0[(16 * mt + j) * 32 + 8 * g + 4 * sq + e];
    } else if (i < OFF_AS0) {
      int q = i - OFF_A1;
      int sq = q >> 8, lane = (q >> 2) & 63, e = q & 3;
      int g = lane >> 4, j = lane & 15;
      v = p.w1[j * 64 + 16 * sq + 4 * g + e];
    } else if (i < OFF_AC0) {
      int q = i - OFF_AS0;
      int mt = q >> 8, lane = (q >> 2) & 63, e = q & 3;
      int g = lane >> 4, j = lane & 15;
      int m = 4 * g + e;  // base output neuron feeding this k slot; neuron 0 is the density logit, not a geo feature
      v = m == 0 ? 0.f : p.ws0[(16 * mt + j) * 15 + (m - 1)];
    } else if (i < OFF_AC1) {
      int q = i - OFF_AC0;
      int mt = q >> 8, lane = (q >> 2) & 63, e = q & 3;
      int g = lane >> 4, j = lane & 15;
      int m = 4 * g + e;
      v = m == 0 ? 0.f : p.wc0[(16 * mt + j) * 63 + 16 + (m - 1)];
    } else if (i < OFF_B0) {
      int q = i - OFF_AC1;
      int mt = q >> 10, sq = (q >> 8) & 3, lane = (q >> 2) & 63, e = q & 3;
      int g = lane >> 4, j = lane & 15;
      v = p.wc1[(16 * mt + j) * 64 + 16 * sq + 4 * g + e];
    } else if (i < OFF_B1) {
      v = p.b0[i - OFF_B0];
    } else if (i < OFF_BS0) {
      v = p.b1[i - OFF_B1];
    } else if (i < OFF_BC1) {
      v = p.bs0[i - OFF_BS0];
    } else if (i < OFF_WF) {
      v = p.bc1[i - OFF_BC1];
    } else if (i < OFF_WRGB) {
      int k = i - OFF_WF;
      for (int m = 0; m < 64; ++m) v = fmaf(p.wh[m], R16(p.ws1[m * 64 + k]), v);
    } else if (i < OFF_MISC) {
      v = R16(p.wc2[i - OFF_WRGB]);
    } else if (i < OFF_WSH) {
      int k = i - OFF_MISC;
      if (k == 0) {
        v = p.bh[0];
        for (int m = 0; m < 64; ++m) v = fmaf(p.wh[m], p.bs1[m], v);
      } else if (k <= 3) {
        v = p.bc2[k - 1];
      }
    } else if (i < OFF_SCALE) {
      int q = i - OFF_WSH;
      v = R16(p.wc0[(q >> 4) * 63 + (q & 15)]);
    } else if (i < OFF_LVL) {
      int q = i - OFF_SCALE;  // static selects: a dynamically indexed kernarg array would go to scratch
      v = p.grid.scale[0];
#pragma unroll
      for (int k = 1; k < CN_MAX_LEVELS; ++k) v = q == k ? p.grid.scale[k] : v;
    } else if (i < BLOB_FLOATS) {
      const int q = (i - OFF_LVL) >> 2, f = (i - OFF_LVL) & 3;
      unsigned u = 0u;
#pragma unroll
      for (int k = 0; k < CN_MAX_LEVELS; ++k) {
        const unsigned w = f == 0 ? p.grid.off[k] : (f == 1 ? p.grid.mask[k] : (f == 2 ? p.grid.m1[k] : p.grid.m2[k]));
        u = q == k ? w : u;
      }
      v = __builtin_bit_cast(float, u);
    } else {
      int q = i - BLOB_FLOATS;
      int row = q >> 6, n = q & 63;
      v = p.bc0[n];
      if (p.app_mode != CN_APP_ZEROS) {
        const float* a = p.app_mode == CN_APP_PER_CAMERA ? p.emb + row * 32 : p.emb_mean;
        for (int k = 0; k < 32; ++k) v = fmaf(R16(p.wc0[n * 63 + 31 + k]), R16(a[k]), v);
      }
      app_bias[q] = v;
      continue;
    }
    blob[i] = v;
  }
}

struct FusedArgs {
  GridDev grid;
  SceneDev scene;
  const float* blob;
  const float* blob_ext;  // split-bf16 mode: the image blocks beyond the fp32 region
  const float* app_bias;  // [rows][64]
  const float* origins;
  const float* directions;
  const float* nears;
  const float* fars;
  const int64_t* cam_idx;
  const float* bins;  // [R,S+1] or null
  long long num_rays;
  int S;
  int spacing;
  int bg_mode;
  float bg[3];
  int app_per_camera;
  int sh_unit;
  int eval_clamp;
  float early_stop;  // 0 = off
  int image_width;        // > 0: rays are pixels [pixel_start, pixel_start + num_rays) of a row-major image
  int stripes_per_xcd;    // column stripes each XCD sweeps one after the other (stripe ~24 pixels wide)
  long long pixel_start;
  // composited outputs
  float *out_rgb, *out_acc, *out_depth, *out_sem, *out_cmap, *out_w;
  // per-sample outputs
  float *s_density, *s_rgb, *s_sem, *s_pos;
  int64_t* s_label;
};

#if CN_ABLATE_MLP  // timing-only build: one VALU op per MFMA keeps every value alive without the matrix pipe
#define MFMA(a, b, c) ((c) + (a) * (b))
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// ReLU as a signed-integer max on the float bits: one v_max_i32 per value.  fmaxf(x, 0) costs two VALU ops here
// (hipcc inserts a canonicalising v_max_f32 x, x in front); on the bit pattern, every negative float (sign bit set,
// -0.0 included) is a negative int -> 0, every non-negative float is unchanged.
__device__ __forceinline__ float relu1(float x) {
  int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  f32x4 r;
  r.x = relu1(v.x);
  r.y = relu1(v.y);
  r.z = relu1(v.z);
  r.w = relu1(v.w);
  return r;
}

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}

__device__ __forceinline__ float group_sum(float v) { return rows_sum(v); }  // lanes j, j+16, j+32, j+48

__device__ __forceinline__ float pick4(int g, float a, float b, float c, float d) {
  return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d));
}

// Level record of a lane (lane group g owns levels 4g..4g+3).  GENERIC = false (nerfstudio torch layout: every level
// hashed with the same primes and mask, level l at l * T): compile-time multipliers, scalar mask -- the code of the
// round-1 kernels.  GENERIC = true (tcnn layout: dense and hashed levels mixed): the record comes from the LDS blob.
template <bool GENERIC>
__device__ __forceinline__ Lvl lane_level_rec(const float* lvl_records, const GridDev& grid, int level, float scale) {
  Lvl lv;
  if constexpr (GENERIC) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 r = *reinterpret_cast<const u32x4*>(lvl_records + 4 * level);
    lv.off = r.x;
    lv.mask = r.y;
    lv.m1 = r.z;
    lv.m2 = r.w;
  } else {
    lv.mask = grid.mask[0];
    lv.off = (unsigned)level * (grid.mask[0] + 1u);
    lv.m1 = CN_P1;
    lv.m2 = CN_P2;
  }
  lv.scale = scale;
  return lv;
}

// HALF: the hash table holds half2 entries (CN_TABLE_F16: 4-byte gathers, 512 algorithmic bytes per sample)
template <bool PER_SAMPLE, bool DENSITY_ONLY, bool HALF = false, bool GENERIC = false>
__global__ void __launch_bounds__(FUSED_THREADS, CN_FUSED_MIN_WAVES_PER_SIMD) render_fused_kernel(FusedArgs A) {
  __shared__ __align__(16) float lds[BLOB_FLOATS + FUSED_WAVES * WAVE_SCRATCH];
  {
    const float4* src = reinterpret_cast<const float4*>(A.blob);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < BLOB_FLOATS / 4; i += FUSED_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int g = lane >> 4, j = lane & 15;
  float* scratch = lds + BLOB_FLOATS + wave * WAVE_SCRATCH;  // [0,64): per-ray colour bias, [64,129): chunk bin edges
  float* tbuf = scratch + 64;
  const int S = A.S;

  // XCD-aware ray ownership: blocks b, b+8, ... share an XCD (round-robin dispatch) and therefore an L2.
  //  * unknown ray order: each XCD sweeps one contiguous eighth of the batch;
  //  * pixel runs of a row-major image (image_width hint): each XCD sweeps COLUMN STRIPES of the image rows the batch
  //    touches, so the ~400 rays it has in flight form a 2-D patch (vertical neighbours share grid cells too).
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const long long stride = (long long)(gridDim.x >> 3) * FUSED_WAVES;
  const bool striped = A.image_width > 0;
  const long long per_xcd = (A.num_rays + 7) >> 3;
  const int nstripe = 8 * A.stripes_per_xcd;
  const int cw = striped ? (A.image_width + nstripe - 1) / nstripe : 0;  // stripe width
  const long long first_row = striped ? A.pixel_start / A.image_width : 0;
  const long long last_row = striped ? (A.pixel_start + A.num_rays - 1) / A.image_width : 0;
  const long long rows = last_row - first_row + 1;
  const long long items =
      striped ? rows * cw * A.stripes_per_xcd : min(per_xcd, max(A.num_rays - xcd * per_xcd, 0LL));

  for (long long q = slot * FUSED_WAVES + wave; q < items; q += stride) {
    long long rr;
    if (striped) {
      const long long sq = q / (rows * cw);  // which of this XCD's stripes (swept one after the other)
      const long long qq = q - sq * rows * cw;
      const long long vrow = qq / cw;
      const int col = (int)(sq * 8 + xcd) * cw + (int)(qq - vrow * cw);
      rr = (first_row + vrow) * A.image_width + col - A.pixel_start;
      if (col >= A.image_width || rr < 0 || rr >= A.num_rays) continue;  // wave-uniform
    } else {
      rr = xcd * per_xcd + q;
    }
    const long long r = __builtin_amdgcn_readfirstlane((int)rr);  // wave-uniform -> scalar loads below
    const float ox = A.origins[3 * r], oy = A.origins[3 * r + 1], oz = A.origins[3 * r + 2];
    const float dx = A.directions[3 * r], dy = A.directions[3 * r + 1], dz = A.directions[3 * r + 2];
    const float near = A.nears[r], far = A.fars[r];
    const float sn = spacing_fn(A.spacing, near), sf = spacing_fn(A.spacing, far);
    const float* bins = A.bins ? A.bins + r * (long long)(S + 1) : nullptr;
    auto edge = [&](int i) -> float {
      i = min(i, S);
      return bins ? bins[i] : spacing_to_euclid(A.spacing, linspace01(i, S + 1), sn, sf);
    };

    // ---- per-ray colour bias: bc0 + Wc0[:, sh].SH(d) + Wc0[:, app].app  (lane n = neuron n) ------------------
    if (!DENSITY_ONLY) {
      float sx = dx, sy = dy, sz = dz;
      if (!A.sh_unit) {
        sx = (dx + 1.f) / 2.f;
        sy = (dy + 1.f) / 2.f;
        sz = (dz + 1.f) / 2.f;
      }
      float sh[16];
      sh_deg4(sx, sy, sz, sh);
      long long row = A.app_per_camera ? A.cam_idx[r] : 0;
      float bias = A.app_bias[row * 64 + lane];
      const f32x4* wsh = reinterpret_cast<const f32x4*>(lds + OFF_WSH + lane * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 w = wsh[q];
        bias = fmaf(w.x, sh[4 * q + 0], bias);
        bias = fmaf(w.y, sh[4 * q + 1], bias);
        bias = fmaf(w.z, sh[4 * q + 2], bias);
        bias = fmaf(w.w, sh[4 * q + 3], bias);
      }
      __builtin_amdgcn_wave_barrier();
      scratch[lane] = bias;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    CompositeState st;
    for (int c0 = 0; c0 < S; c0 += 64) {
      // ---- bin edges of the chunk: lane l owns edge c0+l, edge c0+64 is wave-uniform; staged in LDS so that the
      //      gather lanes (sample 32h+16c+j) and the compositing lanes (sample l) read the same values ----------
      const float e_lo = edge(c0 + lane);
      const float e_top = edge(c0 + 64);
      __builtin_amdgcn_wave_barrier();
      tbuf[lane] = e_lo;
      if (lane == 0) tbuf[64] = e_top;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

      // values of "my" sample (lane l <-> sample c0+l), filled by the half that evaluates it
      float my_dlogit = 0.f, my_sel = 0.f, my_sem = 0.f, my_r = 0.f, my_g = 0.f, my_b = 0.f;

      // The 64 samples go through gather + MLPs as two halves of 2 column tiles (32 samples): half the live
      // accumulators (h: 32, c1: 32 VGPRs) buys twice the waves per SIMD to hide the gather latency.
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        // ---- positions of this lane's two gather samples (rows past S sit at the far plane and are discarded) --
        float px[2], py[2], pz[2];
        bool sel[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int k = 32 * half + 16 * c + j;
          float mid = (tbuf[k] + tbuf[k + 1]) / 2.f;
          px[c] = ox + dx * mid;
          py[c] = oy + dy * mid;
          pz[c] = oz + dz * mid;
          sel[c] = normalize_position(A.scene, px[c], py[c], pz[c]);
        }
        // ---- hash grid: levels 4g..4g+3 for the two samples ------------------------------------------------
        f32x4 feat[2][2];
        {
          const f32x4 lvl_scale = *reinterpret_cast<const f32x4*>(lds + OFF_SCALE + 4 * g);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const Lvl lv = lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g + q, lvl_scale[q]);
            const float pos_off = GENERIC ? A.grid.pos_offset : 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#if CN_ABLATE_GATHER  // timing-only build: no table reads (positions still feed the MLP so nothing is dead code)
              float2 f = make_float2(px[c] * lv.scale, py[c] + pz[c]);
#else
              float2 f = hash_level_sc<HALF, GENERIC>(A.grid.table, lv, pos_off, px[c], py[c], pz[c]);
#endif
              if (q == 0) { feat[c][0].x = f.x; feat[c][0].y = f.y; }
              if (q == 1) { feat[c][0].z = f.x; feat[c][0].w = f.y; }
              if (q == 2) { feat[c][1].x = f.x; feat[c][1].y = f.y; }
              if (q == 3) { feat[c][1].z = f.x; feat[c][1].w = f.y; }
            }
            __builtin_amdgcn_sched_barrier(0);  // one level (2 samples) of gathers in flight at a time
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- base MLP layer 0: 32 -> 64, ReLU ----------------------------------------------------------------
        f32x4 h[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_B0 + 16 * mt + 4 * g);
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(lds + OFF_A0 + ((mt * 2 + 0) * 64 + lane) * 4);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(lds + OFF_A0 + ((mt * 2 + 1) * 64 + lane) * 4);
          f32x4 acc[2] = {b, b};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = MFMA(a0[e], feat[c][0][e], acc[c]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = MFMA(a1[e], feat[c][1][e], acc[c]);
#pragma unroll
          for (int c = 0; c < 2; ++c) h[mt][c] = relu4(acc[c]);
        }
        // ---- base MLP layer 1: 64 -> 16 (neuron 0 = density logit, 1..15 = geo features) -------------------------
        f32x4 o16[2];
        {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + 4 * g);
          f32x4 acc[2] = {b, b};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_A1 + (t * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], h[t][c][e], acc[c]);
          }
          o16[0] = acc[0];
          o16[1] = acc[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool mine = (g >> 1) == half;  // this half holds lane l's own sample in column tile g&1
        const bool odd = (g & 1) != 0;
        {
          float d0 = row0_broadcast(o16[0].x), d1 = row0_broadcast(o16[1].x);
          float dsel = odd ? d1 : d0;
          float ssel = (odd ? sel[1] : sel[0]) ? 1.f : 0.f;
          my_dlogit = mine ? dsel : my_dlogit;
          my_sel = mine ? ssel : my_sel;
        }
        if (!DENSITY_ONLY) {
          // ---- semantics: relu(Ws0 geo + bs0) . (Wh Ws1) + folded bias ------------------------------------------
          float sem_part[2] = {0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_BS0 + 16 * mt + 4 * g);
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_AS0 + (mt * 64 + lane) * 4);
            const f32x4 wf = *reinterpret_cast<const f32x4*>(lds + OFF_WF + 16 * mt + 4 * g);
            f32x4 acc[2] = {b, b};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], o16[c][e], acc[c]);
#pragma unroll
            for (int c = 0; c < 2; ++c) sem_part[c] = dot4(wf, relu4(acc[c]), sem_part[c]);
          }
          // ---- colour layer 0: geo columns on the MFMA, SH + appearance columns pre-summed in the ray bias -------
          f32x4 c1[4][2];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_AC0 + (mt * 64 + lane) * 4);
            const f32x4 cb = *reinterpret_cast<const f32x4*>(scratch + 16 * mt + 4 * g);
            f32x4 acc[2] = {cb, cb};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], o16[c][e], acc[c]);
#pragma unroll
            for (int c = 0; c < 2; ++c) c1[mt][c] = relu4(acc[c]);
          }
          __builtin_amdgcn_sched_barrier(0);
          // ---- colour layer 1 (64 -> 64, ReLU) with the 64 -> 3 head folded into the row-tile loop ----------------
          float rgb_part[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_BC1 + 16 * mt + 4 * g);
            f32x4 acc[2] = {b, b};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_AC1 + ((mt * 4 + t) * 64 + lane) * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], c1[t][c][e], acc[c]);
            }
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds + OFF_WRGB + 0 * 64 + 16 * mt + 4 * g);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(lds + OFF_WRGB + 1 * 64 + 16 * mt + 4 * g);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds + OFF_WRGB + 2 * 64 + 16 * mt + 4 * g);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              f32x4 v = relu4(acc[c]);
              rgb_part[c][0] = dot4(w0, v, rgb_part[c][0]);
              rgb_part[c][1] = dot4(w1, v, rgb_part[c][1]);
              rgb_part[c][2] = dot4(w2, v, rgb_part[c][2]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          // ---- reduce the heads over the four lane groups and keep the column tile this lane owns -------------
          float s0 = group_sum(sem_part[0]), s1 = group_sum(sem_part[1]);
          my_sem = mine ? (odd ? s1 : s0) : my_sem;
          float r0 = group_sum(rgb_part[0][0]), r1 = group_sum(rgb_part[1][0]);
          my_r = mine ? (odd ? r1 : r0) : my_r;
          float g0 = group_sum(rgb_part[0][1]), g1 = group_sum(rgb_part[1][1]);
          my_g = mine ? (odd ? g1 : g0) : my_g;
          float b0 = group_sum(rgb_part[0][2]), b1 = group_sum(rgb_part[1][2]);
          my_b = mine ? (odd ? b1 : b0) : my_b;
        }
      }
      // ---- lane l now holds sample c0 + l ------------------------------------------------------------------
      float density = expf(my_dlogit) * my_sel;
      float sem = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
      if (!DENSITY_ONLY) {
        sem = my_sem + lds[OFF_MISC + 0];
        cr = sigmoidf(my_r + lds[OFF_MISC + 1]);
        cg = sigmoidf(my_g + lds[OFF_MISC + 2]);
        cb = sigmoidf(my_b + lds[OFF_MISC + 3]);
      }
      const int i = c0 + lane;
      const bool valid = i < S;
      const float e0 = e_lo, e1 = tbuf[lane + 1];
      const float mid = (e0 + e1) / 2.f;
      if (PER_SAMPLE) {
        if (valid) {
          const long long o = r * (long long)S + i;
          if (A.s_density) A.s_density[o] = density;
          if (A.s_sem) A.s_sem[o] = sem;
          if (A.s_label) A.s_label[o] = (int64_t)semantics_label(sem);
          if (A.s_rgb) {
            A.s_rgb[3 * o + 0] = cr;
            A.s_rgb[3 * o + 1] = cg;
            A.s_rgb[3 * o + 2] = cb;
          }
          if (A.s_pos) {
            A.s_pos[3 * o + 0] = ox + dx * mid;
            A.s_pos[3 * o + 1] = oy + dy * mid;
            A.s_pos[3 * o + 2] = oz + dz * mid;
          }
        }
      } else {
        float w = composite_chunk(st, valid, i == S - 1, e1 - e0, density, mid, cr, cg, cb, sem, A.eval_clamp != 0);
        if (A.out_w && valid) A.out_w[r * (long long)S + i] = w;
        // Optional early ray termination (cn_render_opts.early_stop_transmittance; the reference has none, so it is off
        // by default): once the transmittance behind this chunk is below the threshold the remaining samples carry
        // less than that much weight in total -- the wave drops them and moves on to its next ray.  The "last sample"
        // background takes the last evaluated sample's colour (it is scaled by 1 - accumulation < threshold).
        if (A.early_stop > 0.f && c0 + 64 < S && __expf(-st.carry_dd) < A.early_stop) {  // wave-uniform
          st.last_r = wave_read(A.eval_clamp ? nan_to_num(cr) : cr, 63);
          st.last_g = wave_read(A.eval_clamp ? nan_to_num(cg) : cg, 63);
          st.last_b = wave_read(A.eval_clamp ? nan_to_num(cb) : cb, 63);
          st.last_mid = wave_read(mid, 63);
          if (A.out_w)
            for (int k = c0 + 64 + lane; k < S; k += 64) A.out_w[r * (long long)S + k] = 0.f;
          break;
        }
      }
    }
    if (!PER_SAMPLE) {
      CompositeOut o = composite_finish(st, A.bg_mode, A.bg[0], A.bg[1], A.bg[2], A.eval_clamp != 0);
      if (lane == 0) {
        if (A.out_acc) A.out_acc[r] = o.acc;
        if (A.out_depth) A.out_depth[r] = o.depth;
        if (!DENSITY_ONLY) {
          if (A.out_rgb) {
            A.out_rgb[3 * r + 0] = o.r;
            A.out_rgb[3 * r + 1] = o.g;
            A.out_rgb[3 * r + 2] = o.b;
          }
          if (A.out_sem) A.out_sem[r] = o.sem;
          if (A.out_cmap) {
            float l = semantics_label(o.sem);
            A.out_cmap[3 * r + 0] = l;
            A.out_cmap[3 * r + 1] = l;
            A.out_cmap[3 * r + 2] = l;
          }
        }
      }
    }
  }
}

}  // namespace cn
#include "render_split.hpp"
#include "render_f16.hpp"
namespace cn {

int validate_field(const cn_field_params& p);  // field_simple.hip

static int check_fused_shape(const cn_field_params& p) {
  int rc = validate_field(p);
  if (rc) return rc;
  bool ok = p.grid.num_levels == 16 && p.geo_feat_dim == 15 && p.app_dim == 32 && p.base.num_layers == 2 &&
            p.base.dims[0] == 32 && p.base.dims[1] == 64 && p.base.dims[2] == 16 && p.semantics.num_layers == 2 &&
            p.semantics.dims[0] == 15 && p.semantics.dims[1] == 64 && p.semantics.dims[2] == 64 &&
            p.color.num_layers == 3 && p.color.dims[0] == 63 && p.color.dims[1] == 64 && p.color.dims[2] == 64 &&
            p.color.dims[3] == 3;
  CN_REQUIRE(ok, CN_ERR_UNSUPPORTED,
             "fused renderer is built for the default fruit_nerf_method field shape "
             "(16 levels, 32->64->16, 15->64->64->1, 63->64->64->3, appearance 32); use cn_field_eval + cn_composite");
  return CN_OK;
}

static size_t fused_workspace_bytes(const cn_field_params* p) {
  int rows = p && p->num_images > 0 ? p->num_images : 1;
  // weight image | appearance bias rows | mean | split-bf16 image extension
  return (size_t)(BLOB_FLOATS + rows * 64 + 32 + BF16_EXT_FLOATS) * sizeof(float);
}

// Per-device launch geometry of the render kernels, and the dynamic-LDS attribute of the producer/consumer variants.
struct FusedDevice {
  int split_resident;  // workgroups of render_split_kernel the device holds at once (a multiple of 8 = XCD teams)
  int res_sample, res_density, res_full;  // resident blocks of the render_fused_kernel variants
  int res_f16_sample, res_f16_density, res_f16_full;  // ... of the render_f16_kernel variants
};

template <typename K>
static hipError_t resident_blocks(K kernel, int cus, int* out) {
  // Resident blocks of a kernel variant (rounded down to a multiple of 8 = one group per XCD).  A grid larger than the
  // residency would leave a tail running at a fraction of the occupancy.
  int per_cu = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, FUSED_THREADS, 0);
  if (e != hipSuccess) return e;
  if (per_cu < 1) per_cu = 1;
  const int n = cus * per_cu;
  *out = n >= 8 ? (n / 8) * 8 : 8;
  return hipSuccess;
}

template <bool PS, int MM, bool H>
static hipError_t split_attr() {
  const int bytes = (int)(MM == MM_BF16 ? SPLIT_LDS_BYTES_BF16 : SPLIT_LDS_BYTES);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(render_split_kernel<PS, MM, H, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(render_split_kernel<PS, MM, H, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if constexpr (MM == MM_FP32) {  // the packed schedule of 48-sample rays (render_split.hpp: PACK)
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(render_split_kernel<PS, MM, H, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPLIT_LDS_BYTES_PACK);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(render_split_kernel<PS, MM, H, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPLIT_LDS_BYTES_PACK);
  }
  return e;
}

static hipError_t fused_device_init(int dev, FusedDevice& d) {
  int cus = 256;
  hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
#define CN_TRY(x) if ((e = (x)) != hipSuccess) return e
  CN_TRY((split_attr<false, MM_FP32, false>()));
  CN_TRY((split_attr<true, MM_FP32, false>()));
  CN_TRY((split_attr<false, MM_BF16, false>()));
  CN_TRY((split_attr<true, MM_BF16, false>()));
  CN_TRY((split_attr<false, MM_F16, false>()));
  CN_TRY((split_attr<true, MM_F16, false>()));
  CN_TRY((split_attr<false, MM_FP32, true>()));
  CN_TRY((split_attr<true, MM_FP32, true>()));
  CN_TRY((split_attr<false, MM_BF16, true>()));
  CN_TRY((split_attr<true, MM_BF16, true>()));
  CN_TRY((split_attr<false, MM_F16, true>()));
  CN_TRY((split_attr<true, MM_F16, true>()));
  int per_cu = 0;
  CN_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_split_kernel<false, MM_FP32, false, false>,
                                                      SPLIT_THREADS, SPLIT_LDS_BYTES));
  if (per_cu < 1) per_cu = 1;
  const int r = cus * per_cu;
  d.split_resident = r >= 8 ? (r / 8) * 8 : 8;
  CN_TRY(resident_blocks(render_fused_kernel<true, false, false, false>, cus, &d.res_sample));
  CN_TRY(resident_blocks(render_fused_kernel<false, true, false, false>, cus, &d.res_density));
  CN_TRY(resident_blocks(render_fused_kernel<false, false, false, false>, cus, &d.res_full));
  CN_TRY(resident_blocks(render_f16_kernel<true, false, true, true>, cus, &d.res_f16_sample));
  CN_TRY(resident_blocks(render_f16_kernel<false, true, true, true>, cus, &d.res_f16_density));
  CN_TRY(resident_blocks(render_f16_kernel<false, false, true, true>, cus, &d.res_f16_full));
#undef CN_TRY
  return hipSuccess;
}

static PerDevice<FusedDevice> g_fused_devices;

template <bool PER_SAMPLE>
static int launch_fused(const cn_field_params* params, const cn_scene* scene, const cn_render_opts* opts,
                        const float* origins, const float* directions, const float* nears, const float* fars,
                        const int64_t* cam_idx, const float* bins, int64_t num_rays, FusedArgs& A, void* workspace,
                        size_t workspace_bytes, cn_stream_t stream, const char* who) {
  CN_REQUIRE(params && scene && opts, CN_ERR_INVALID, "%s: null parameter struct", who);
  if (num_rays <= 0) return CN_OK;  // an empty batch is a no-op (empty tensors carry null data pointers)
  CN_REQUIRE(origins && directions && nears && fars, CN_ERR_INVALID, "%s: null input", who);
  CN_REQUIRE(opts->num_samples > 0, CN_ERR_INVALID, "%s: num_samples must be > 0", who);
  CN_REQUIRE(opts->spacing == CN_SPACING_UNIFORM || opts->spacing == CN_SPACING_PIECEWISE, CN_ERR_INVALID,
             "%s: unknown spacing %d", who, opts->spacing);
  CN_REQUIRE(opts->app_mode >= CN_APP_ZEROS && opts->app_mode <= CN_APP_PER_CAMERA, CN_ERR_INVALID, "%s: app_mode %d",
             who, opts->app_mode);
  CN_REQUIRE(opts->app_mode != CN_APP_PER_CAMERA || cam_idx, CN_ERR_INVALID, "Camera indices are not provided.");
  CN_REQUIRE(opts->bg_mode == CN_BG_LAST_SAMPLE || opts->bg_mode == CN_BG_COLOR, CN_ERR_INVALID, "%s: bg_mode %d", who,
             opts->bg_mode);
  CN_REQUIRE(opts->image_width <= 0 || opts->pixel_start >= 0, CN_ERR_INVALID, "%s: negative pixel_start", who);
  CN_REQUIRE(opts->matrix_precision == CN_MATRIX_FP32 || opts->matrix_precision == CN_MATRIX_SPLIT_BF16 ||
                 opts->matrix_precision == CN_MATRIX_F16,
             CN_ERR_INVALID, "%s: matrix_precision %d", who, opts->matrix_precision);
  int rc = check_fused_shape(*params);
  if (rc) return rc;
  rc = check_grid(params->grid, false, who);
  if (rc) return rc;
  CN_REQUIRE(workspace && workspace_bytes >= fused_workspace_bytes(params), CN_ERR_WORKSPACE,
             "%s: workspace %zu B < %zu B", who, workspace_bytes, fused_workspace_bytes(params));
  CN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, CN_ERR_INVALID, "%s: workspace must be 16-B aligned",
             who);
  CN_REQUIRE(num_rays < (1LL << 31), CN_ERR_INVALID, "%s: at most 2^31-1 rays per call", who);
  const FusedDevice* dev = nullptr;
  rc = g_fused_devices.get(fused_device_init, &dev, who);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  float* blob = static_cast<float*>(workspace);
  float* app_bias = blob + BLOB_FLOATS;
  // ---- which kernel: the producer/consumer variant (render_split.hpp) or the single-wave one -----------------------------
  // (the environment is read per call so that one process can compare both forms)
  const char* split_env = getenv("CN_FUSED_SPLIT");
  const int split_mode = split_env ? atoi(split_env) : CN_FUSED_SPLIT_DEFAULT;
  const float early_stop = opts->early_stop_transmittance > 0.f ? opts->early_stop_transmittance : 0.f;
  unsigned split_blocks = 0;  // 0: render_fused_kernel
  // Early termination: both kernels implement it.  The workgroup of the split kernel runs its 8 rays in lock-step, so
  // it only saves time when they all go opaque at a similar depth (C2 batch on an opaque medium: 2.53 -> 1.13 ms); the
  // independent waves of the fused kernel save in proportion to the terminated rays (3.2 -> 0.87 ms), which suits real
  // scenes (a mix of rays that hit a surface and rays that cross empty space) better -- so it is the default there and
  // CN_FUSED_SPLIT=2 forces the split kernel.
  // fp16 matrix mode: the composited and per-sample renders run the fp16 form of the producer/consumer kernel
  // (render_split_kernel<., MM_F16, ., .>) at EVERY batch size, so that a ray's result does not depend on the call it is part
  // of; the density-only pass, which the split kernel does not have, runs render_f16_kernel (render_f16.hpp).
  // CN_F16_KERNEL=own is the A/B switch that sends all three through render_f16_kernel (2.9 vs 1.6 ms at C2).
  const bool want_f16 = opts->matrix_precision == CN_MATRIX_F16;
  const char* f16_env = getenv("CN_F16_KERNEL");
  const bool f16_split = want_f16 && !opts->density_only && !(f16_env && strcmp(f16_env, "own") == 0);
  const bool f16_own = want_f16 && !f16_split;
  if (f16_split || (!want_f16 && !opts->density_only && split_mode && (PER_SAMPLE || early_stop == 0.f || split_mode > 1))) {
    const int resident = dev->split_resident;
    const long long work = PER_SAMPLE ? num_rays * ((opts->num_samples + 63) / 64) : num_rays;  // (ray, chunk) items
    const long long want_s = (((work + SPLIT_PAIRS - 1) / SPLIT_PAIRS) + 7) / 8 * 8;
    // one workgroup per CU carries 8 rays at a time: with fewer rays than that fills the device (the exporters' 512-ray
    // x 3000-sample calls) the 4-wave workgroups of render_fused_kernel spread over more CUs
    if (want_s >= resident || split_mode > 1 || f16_split) split_blocks = (unsigned)(want_s < resident ? want_s : resident);
  }
  // split-bf16 matrix products: an option of the producer/consumer kernel only; elsewhere the products stay fp32
  const int mm = (split_blocks > 0 || f16_own) ? opts->matrix_precision : CN_MATRIX_FP32;
  const bool bf16 = mm == MM_BF16;
  A.grid = make_grid_dev(params->grid);
  const bool half = A.grid.half != 0;
  const bool generic = params->grid.layout != CN_GRID_TORCH;
  PrepArgs P;
  P.mm = mm;
  P.ext = blob + BLOB_FLOATS + (size_t)(params->num_images > 0 ? params->num_images : 1) * 64 + 32;
  P.w0 = params->base.weight[0];
  P.b0 = params->base.bias[0];
  P.w1 = params->base.weight[1];
  P.b1 = params->base.bias[1];
  P.ws0 = params->semantics.weight[0];
  P.bs0 = params->semantics.bias[0];
  P.ws1 = params->semantics.weight[1];
  P.bs1 = params->semantics.bias[1];
  P.wh = params->sem_head_weight;
  P.bh = params->sem_head_bias;
  P.wc0 = params->color.weight[0];
  P.bc0 = params->color.bias[0];
  P.wc1 = params->color.weight[1];
  P.bc1 = params->color.bias[1];
  P.wc2 = params->color.weight[2];
  P.bc2 = params->color.bias[2];
  P.emb = params->appearance;
  P.num_images = params->num_images;
  P.app_mode = opts->app_mode;
  P.app_rows = opts->app_mode == CN_APP_PER_CAMERA ? params->num_images : 1;
  float* emb_mean = app_bias + (size_t)(params->num_images > 0 ? params->num_images : 1) * 64;
  P.emb_mean = emb_mean;
  if (opts->app_mode == CN_APP_MEAN)
    hipLaunchKernelGGL(prep_mean_kernel, dim3(1), dim3(256), 0, s, params->appearance, params->num_images, emb_mean);
  P.grid = A.grid;
  hipLaunchKernelGGL(prep_kernel, dim3(48), dim3(256), 0, s, P, blob, app_bias);
  rc = check_launch("cn_render prep");
  if (rc) return rc;

  A.scene = make_scene_dev(*scene);
  A.blob = blob;
  A.blob_ext = P.ext;
  A.app_bias = app_bias;
  A.origins = origins;
  A.directions = directions;
  A.nears = nears;
  A.fars = fars;
  A.cam_idx = cam_idx;
  A.bins = bins;
  A.num_rays = num_rays;
  A.S = opts->num_samples;
  A.spacing = opts->spacing;
  A.bg_mode = opts->bg_mode;
  A.bg[0] = opts->bg_color[0];
  A.bg[1] = opts->bg_color[1];
  A.bg[2] = opts->bg_color[2];
  A.app_per_camera = opts->app_mode == CN_APP_PER_CAMERA;
  A.sh_unit = opts->sh_unit_dir;
  A.eval_clamp = opts->eval_clamp;
  A.early_stop = early_stop;
  A.image_width = opts->image_width > 0 ? opts->image_width : 0;
  // ~400 rays are in flight per XCD; a stripe about 24 pixels wide makes that patch roughly square (measured at
  // 800 px: 1/2/4/8 stripes per XCD -> 3.57 / 3.78 / 3.83 / 3.66 Gsamples/s)
  A.stripes_per_xcd = A.image_width > 0 ? (A.image_width + 96) / 192 : 1;
  if (const char* e = getenv("CN_STRIPES_PER_XCD")) A.stripes_per_xcd = atoi(e);  // tuning aid
  if (A.stripes_per_xcd < 1) A.stripes_per_xcd = 1;
  A.pixel_start = opts->pixel_start;
  // persistent grid: exactly the resident block count (a multiple of 8 = XCD groups), never more waves than rays
  const long long cap = f16_own ? (PER_SAMPLE ? dev->res_f16_sample : (opts->density_only ? dev->res_f16_density : dev->res_f16_full))
                                : (PER_SAMPLE ? dev->res_sample : (opts->density_only ? dev->res_density : dev->res_full));
  const long long fused_items = (f16_own && PER_SAMPLE) ? num_rays * ((opts->num_samples + 63) / 64) : num_rays;
  const long long want = (((fused_items + FUSED_WAVES - 1) / FUSED_WAVES) + 7) / 8 * 8;
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
  if (split_blocks) {
    const size_t lds_bytes = bf16 ? SPLIT_LDS_BYTES_BF16 : SPLIT_LDS_BYTES;
    const char* pack_env = getenv("CN_SPLIT_PACK");
    const bool pack = mm == MM_FP32 && opts->num_samples > 32 && opts->num_samples <= 48 && early_stop == 0.f &&
                      !(pack_env && atoi(pack_env) == 0);
#define CN_SPLIT_LAUNCH(MM, H)                                                                                       \
  do {                                                                                                               \
    if (generic)                                                                                                     \
      hipLaunchKernelGGL((render_split_kernel<PER_SAMPLE, MM, H, true>), dim3(split_blocks), dim3(SPLIT_THREADS),    \
                         lds_bytes, s, A);                                                                           \
    else                                                                                                             \
      hipLaunchKernelGGL((render_split_kernel<PER_SAMPLE, MM, H, false>), dim3(split_blocks), dim3(SPLIT_THREADS),   \
                         lds_bytes, s, A);                                                                           \
  } while (0)
    if (mm == MM_F16 && half) CN_SPLIT_LAUNCH(MM_F16, true);
    else if (mm == MM_F16) CN_SPLIT_LAUNCH(MM_F16, false);
    else if (bf16 && half) CN_SPLIT_LAUNCH(MM_BF16, true);
    else if (bf16) CN_SPLIT_LAUNCH(MM_BF16, false);
    else if (pack) {
      // 32 < S <= 48 in exact fp32 -- the default method's 48 field samples per ray --: two rays in three half-steps instead of
      // four (render_split.hpp: PACK).  CN_SPLIT_PACK=0 keeps the one-ray schedule (A/B runs, the bit-identity test).
#define CN_SPLIT_LAUNCH_PACK(H, G)                                                                                  \
  hipLaunchKernelGGL((render_split_kernel<PER_SAMPLE, MM_FP32, H, G, true>), dim3(split_blocks), dim3(SPLIT_THREADS), \
                     SPLIT_LDS_BYTES_PACK, s, A)
      if (half && generic) CN_SPLIT_LAUNCH_PACK(true, true);
      else if (half) CN_SPLIT_LAUNCH_PACK(true, false);
      else if (generic) CN_SPLIT_LAUNCH_PACK(false, true);
      else CN_SPLIT_LAUNCH_PACK(false, false);
#undef CN_SPLIT_LAUNCH_PACK
    }
    else if (half) CN_SPLIT_LAUNCH(MM_FP32, true);
    else CN_SPLIT_LAUNCH(MM_FP32, false);
#undef CN_SPLIT_LAUNCH
    return check_launch(who);
  }
  if (f16_own) {
#define CN_F16_LAUNCH1(PS, DO, H)                                                                                  \
  do {                                                                                                             \
    if (generic) hipLaunchKernelGGL((render_f16_kernel<PS, DO, H, true>), dim3(blocks), dim3(FUSED_THREADS), 0, s, A);   \
    else hipLaunchKernelGGL((render_f16_kernel<PS, DO, H, false>), dim3(blocks), dim3(FUSED_THREADS), 0, s, A);          \
  } while (0)
#define CN_F16_LAUNCH(PS, DO)              \
  do {                                     \
    if (half) CN_F16_LAUNCH1(PS, DO, true);  \
    else CN_F16_LAUNCH1(PS, DO, false);      \
  } while (0)
    if (PER_SAMPLE) CN_F16_LAUNCH(true, false);
    else if (opts->density_only) CN_F16_LAUNCH(false, true);
    else CN_F16_LAUNCH(false, false);
#undef CN_F16_LAUNCH
#undef CN_F16_LAUNCH1
    return check_launch(who);
  }
#define CN_FUSED_LAUNCH1(PS, DO, H)                                                                                \
  do {                                                                                                             \
    if (generic) hipLaunchKernelGGL((render_fused_kernel<PS, DO, H, true>), dim3(blocks), dim3(FUSED_THREADS), 0, s, A);  \
    else hipLaunchKernelGGL((render_fused_kernel<PS, DO, H, false>), dim3(blocks), dim3(FUSED_THREADS), 0, s, A);         \
  } while (0)
#define CN_FUSED_LAUNCH(PS, DO)           \
  do {                                    \
    if (half) CN_FUSED_LAUNCH1(PS, DO, true);  \
    else CN_FUSED_LAUNCH1(PS, DO, false);      \
  } while (0)
  if (PER_SAMPLE) CN_FUSED_LAUNCH(true, false);
  else if (opts->density_only) CN_FUSED_LAUNCH(false, true);
  else CN_FUSED_LAUNCH(false, false);
#undef CN_FUSED_LAUNCH
#undef CN_FUSED_LAUNCH1
  return check_launch(who);
}

}  // namespace cn

extern "C" size_t cn_render_workspace_bytes(const cn_field_params* params) { return cn::fused_workspace_bytes(params); }

extern "C" int cn_render_rays(const cn_field_params* params, const cn_scene* scene, const cn_render_opts* opts,
                              const float* origins, const float* directions, const float* nears, const float* fars,
                              const int64_t* camera_indices, const float* bins, int64_t num_rays, float* out_rgb,
                              float* out_accumulation, float* out_depth, float* out_semantics,
                              float* out_semantics_colormap, float* out_weights, void* workspace,
                              size_t workspace_bytes, cn_stream_t stream) {
  cn::FusedArgs A{};
  A.out_rgb = out_rgb;
  A.out_acc = out_accumulation;
  A.out_depth = out_depth;
  A.out_sem = out_semantics;
  A.out_cmap = out_semantics_colormap;
  A.out_w = out_weights;
  return cn::launch_fused<false>(params, scene, opts, origins, directions, nears, fars, camera_indices, bins, num_rays,
                                 A, workspace, workspace_bytes, stream, "cn_render_rays");
}

extern "C" int cn_render_samples(const cn_field_params* params, const cn_scene* scene, const cn_render_opts* opts,
                                 const float* origins, const float* directions, const float* nears, const float* fars,
                                 const int64_t* camera_indices, const float* bins, int64_t num_rays, float* density,
                                 float* rgb, float* semantics, float* positions, int64_t* semantics_colormap,
                                 void* workspace, size_t workspace_bytes, cn_stream_t stream) {
  cn::FusedArgs A{};
  A.s_density = density;
  A.s_rgb = rgb;
  A.s_sem = semantics;
  A.s_pos = positions;
  A.s_label = semantics_colormap;
  return cn::launch_fused<true>(params, scene, opts, origins, directions, nears, fars, camera_indices, bins, num_rays, A,
                                workspace, workspace_bytes, stream, "cn_render_samples");
}

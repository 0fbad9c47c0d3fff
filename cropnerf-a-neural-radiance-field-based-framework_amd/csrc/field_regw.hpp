// FruitField evaluation with the MLP weights RESIDENT IN REGISTERS (included by field_simple.hip).
//
// field_eval_mfma_kernel re-stages every layer's weights from L2 into LDS for each 64-sample tile (185 KB per tile for the
// fruit_nerf_method_big shape: they do not fit LDS next to the activations).  Here the weights never move after the
// prologue: wave w of a 512-thread workgroup owns the 16 output rows [16w, 16w + 16) of every layer that has them and
// keeps those rows -- K / 4 registers per layer, 148 for the big shape -- as MFMA A operands for the whole kernel; the
// activations of a 64-sample tile live in LDS as [feature][68] and are the B operands of all four column tiles.  Layers
// narrower than 128 leave some waves idle (64 rows = 4 waves), the price for never re-reading a weight.
//
// Instantiated for the two field shapes of the reference's method configs:
//   <GEO 15, NS 2, SW 64>   fruit_nerf_method            (base 32-64-16, semantics 15-64-64-1, colour 63-64-64-3)
//   <GEO 30, NS 3, SW 128>  fruit_nerf_method_big/_huge  (base 32-64-31, semantics 30-128-128-64-1, colour 78-64-64-3)
#pragma once

namespace cn {
namespace rw {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TS = 64, LDA = 68, NT = 512, NW = 8;

constexpr int pad16(int n) { return (n + 15) & ~15; }

// A layer with N outputs has NT_ = pad16(N) / 16 row tiles (1, 2, 4 or 8).  The 8 waves form 8 / NT_ groups: wave w owns
// row tile w % NT_ and, of the four 16-sample column tiles, those of its group w / NT_ -- so that every wave has work
// in every layer with at least 32 (row tile, column tile) blocks, at the price of holding the same rows in 8 / NT_ waves.
// A operands of the row tile: lane (i = lane & 15, q = lane >> 4) holds W[16 * rt + i][16 * kb + 4 * q + e].
template <int K>
struct RowTile {
  static constexpr int KP = pad16(K);
  float a[KP / 4];
  __device__ __forceinline__ void load(const float* __restrict__ W, int N, int wave, int lane) {
    const int ntiles = pad16(N) / 16;
    const int i = lane & 15, q = lane >> 4, n = 16 * (wave % ntiles) + i;
#pragma unroll
    for (int j = 0; j < KP / 4; ++j) {
      const int k = 16 * (j >> 2) + 4 * q + (j & 3);
      a[j] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
  }
};

// out = act(b + W in) for this wave's (row tile, column tiles).  Rows >= N come out as zeros (zero weights and bias),
// which is what the next layer's pad inputs must be.
template <int K, bool RELU>
__device__ __forceinline__ void layer(const RowTile<K>& rt, const float* __restrict__ bias, int N, const float* in,
                                      float* out, int wave, int lane) {
  const int ntiles = pad16(N) / 16, groups = NW / ntiles, grp = wave / ntiles, rt0 = 16 * (wave % ntiles);
  // column tiles of this wave's group: 4 / groups each (groups <= 4), or one for the first four groups (groups == 8)
  const int ct_lo = groups <= 4 ? grp * (4 / groups) : grp, ct_hi = groups <= 4 ? ct_lo + 4 / groups : (grp < 4 ? grp + 1 : grp);
  const int i = lane & 15, q = lane >> 4;
  float b4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = rt0 + 4 * q + r;
    b4[r] = n < N ? bias[n] : 0.f;
  }
  // Two column tiles at a time where the wave has two or four: one fp32 matrix instruction is 8 passes and the next one of the
  // SAME accumulation chain cannot start before it ends -- a second, independent chain on the same A operands fills the pipe
  // (two waves per SIMD alone leave it half empty: the kernel ran at 43 % of the fp32 matrix peak).
  if (((ct_hi - ct_lo) & 1) == 0) {
#pragma unroll 1  // (unrolled, hipcc hoists the LDS reads of all four column tiles and spills the resident weights)
    for (int ct = ct_lo; ct < ct_hi; ct += 2) {
      f32x4 acc0 = {b4[0], b4[1], b4[2], b4[3]}, acc1 = acc0;
#pragma unroll
      for (int kb = 0; kb < RowTile<K>::KP / 16; ++kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* row = in + (16 * kb + 4 * q + e) * LDA + 16 * ct + i;
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(rt.a[4 * kb + e], row[0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rt.a[4 * kb + e], row[16], acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v0 = acc0[r], v1 = acc1[r];
        if (RELU) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        out[(rt0 + 4 * q + r) * LDA + 16 * ct + i] = v0;
        out[(rt0 + 4 * q + r) * LDA + 16 * ct + 16 + i] = v1;
      }
    }
    return;
  }
#pragma unroll 1
  for (int ct = ct_lo; ct < ct_hi; ++ct) {
    f32x4 acc = {b4[0], b4[1], b4[2], b4[3]};
    // B operands of round kb + 1 are read from LDS while the four MFMAs of round kb run
    float b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = in[(4 * q + e) * LDA + 16 * ct + i];
#pragma unroll
    for (int kb = 0; kb < RowTile<K>::KP / 16; ++kb) {
      float bn[4];
      if (kb + 1 < RowTile<K>::KP / 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bn[e] = in[(16 * (kb + 1) + 4 * q + e) * LDA + 16 * ct + i];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rt.a[4 * kb + e], b[e], acc, 0, 0, 0);
      if (kb + 1 < RowTile<K>::KP / 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = bn[e];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[r];
      if (RELU) v = fmaxf(v, 0.f);
      out[(rt0 + 4 * q + r) * LDA + 16 * ct + i] = v;
    }
  }
}

template <int GEO, int NS, int SW>
__global__ void __launch_bounds__(NT)
field_eval_regw_kernel(FieldDev fp, SceneDev sc, int app_mode, int sh_unit, const float* __restrict__ origins,
                       const float* __restrict__ directions, const int64_t* __restrict__ cam_idx,
                       const float* __restrict__ starts, const float* __restrict__ ends, long long num_rays, int S,
                       float* __restrict__ density, float* __restrict__ rgb, float* __restrict__ semantics,
                       float* __restrict__ positions) {
  constexpr int H = 64, HT = 64, CW = 64, APP = 32, CIN = 16 + GEO + APP, NG = 1 + GEO;
  extern __shared__ __align__(16) float lds[];
  float* ENC = lds;                          // 32 rows
  float* bufA = ENC + 32 * LDA;              // 128
  float* bufB = bufA + 128 * LDA;            // 128
  float* bufC = bufB + 128 * LDA;            // colour input, pad16(CIN) <= 80
  float* bufG = bufC + 80 * LDA;             // base output: 48 rows, rows >= NG stay zero
  float* scl = bufG + 48 * LDA;              // 16 level scales
  float* app_mean = scl + 16;                // 32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = tid & 63, grp = wave;
  for (int e = tid; e < (32 + 128 + 128 + 80 + 48) * LDA; e += NT) lds[e] = 0.f;
  if (tid < CN_MAX_LEVELS) scl[tid] = fp.grid.scale[tid];
  if (app_mode == CN_APP_MEAN && tid < APP) {
    float m = 0.f;
    for (int n = 0; n < fp.num_images; ++n) m += fp.appearance[(long long)n * APP + tid];
    app_mean[tid] = m / (float)fp.num_images;
  }
  // ---- the weights of this wave's row tiles, for the whole kernel -----------------------------------------------------------
  RowTile<32> w_b0;
  RowTile<H> w_b1;
  RowTile<GEO> w_s0;
  RowTile<SW> w_s1;
  RowTile<SW> w_s2;  // third semantic layer (NS == 3)
  RowTile<CIN> w_c0;
  RowTile<CW> w_c1;
  RowTile<CW> w_c2;
  w_b0.load(fp.base.w[0], H, wave, lane);
  w_b1.load(fp.base.w[1], NG, wave, lane);
  w_s0.load(fp.sem.w[0], SW, wave, lane);
  w_s1.load(fp.sem.w[1], NS == 3 ? SW : HT, wave, lane);
  if (NS == 3) w_s2.load(fp.sem.w[2], HT, wave, lane);
  w_c0.load(fp.color.w[0], CW, wave, lane);
  w_c1.load(fp.color.w[1], CW, wave, lane);
  w_c2.load(fp.color.w[2], 3, wave, lane);
  __syncthreads();

  const long long total = num_rays * (long long)S;
  const long long ntiles = (total + TS - 1) / TS;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long ismp = tile * TS + s;
    const bool valid = ismp < total;
    const long long ic = valid ? ismp : total - 1;
    const long long r = ic / S;
    const float dx = directions[3 * r], dy = directions[3 * r + 1], dz = directions[3 * r + 2];
    const float mid = (starts[ic] + ends[ic]) / 2.f;
    float px = origins[3 * r] + dx * mid, py = origins[3 * r + 1] + dy * mid, pz = origins[3 * r + 2] + dz * mid;
    if (positions && valid && grp == 0) {
      positions[3 * ismp + 0] = px;
      positions[3 * ismp + 1] = py;
      positions[3 * ismp + 2] = pz;
    }
    const bool sel = normalize_position(sc, px, py, pz);
    // ---- inputs: two grid levels per thread; SH and appearance rows of the colour input ------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int l = grp + NW * h;
      const float2 f = hash_level_any(fp.grid, l, px, py, pz);
      ENC[(2 * l) * LDA + s] = f.x;
      ENC[(2 * l + 1) * LDA + s] = f.y;
    }
    if (rgb) {
      if (grp == 1) {
        float sx = dx, sy = dy, sz = dz;
        if (!sh_unit) {
          sx = (dx + 1.f) / 2.f;
          sy = (dy + 1.f) / 2.f;
          sz = (dz + 1.f) / 2.f;
        }
        float sh[16];
        sh_deg4(sx, sy, sz, sh);
#pragma unroll
        for (int k = 0; k < 16; ++k) bufC[k * LDA + s] = sh[k];
      }
      const float* emb = app_mode == CN_APP_PER_CAMERA ? fp.appearance + cam_idx[r] * (long long)APP : nullptr;
#pragma unroll
      for (int h = 0; h < APP / NW; ++h) {
        const int k = grp + NW * h;
        bufC[(16 + GEO + k) * LDA + s] = app_mode == CN_APP_MEAN ? app_mean[k] : (emb ? emb[k] : 0.f);
      }
    }
    __syncthreads();
    layer<32, true>(w_b0, fp.base.b[0], H, ENC, bufA, wave, lane);                 // h1 -> A
    __syncthreads();
    layer<H, false>(w_b1, fp.base.b[1], NG, bufA, bufG, wave, lane);               // logit | geo -> G
    __syncthreads();
    if (grp == 0 && density && valid) density[ismp] = expf(bufG[s]) * (sel ? 1.f : 0.f);
    if (rgb)
      for (int k = grp; k < GEO; k += NW) bufC[(16 + k) * LDA + s] = bufG[(1 + k) * LDA + s];
    layer<GEO, true>(w_s0, fp.sem.b[0], SW, bufG + LDA, bufA, wave, lane);        // s1 -> A
    __syncthreads();
    const float* sout;
    if (NS == 3) {
      layer<SW, true>(w_s1, fp.sem.b[1], SW, bufA, bufB, wave, lane);              // s2 -> B
      __syncthreads();
      layer<SW, false>(w_s2, fp.sem.b[2], HT, bufB, bufA, wave, lane);             // s3 -> A
      sout = bufA;
    } else {
      layer<SW, false>(w_s1, fp.sem.b[1], HT, bufA, bufB, wave, lane);             // s2 -> B
      sout = bufB;
    }
    __syncthreads();
    if (grp == 1 && semantics && valid) {
      float v = fp.sem_head_b[0];
#pragma unroll 8
      for (int k = 0; k < HT; ++k) v = fmaf(fp.sem_head_w[k], sout[k * LDA + s], v);
      semantics[ismp] = v;
    }
    if (rgb) {
      float* c1 = NS == 3 ? bufB : bufA;  // the buffer the semantic output is NOT in
      float* c2 = NS == 3 ? bufA : bufB;
      __syncthreads();                    // (the head above still reads `sout`; c1 is the other buffer, but c2 == sout)
      layer<CIN, true>(w_c0, fp.color.b[0], CW, bufC, c1, wave, lane);
      __syncthreads();
      layer<CW, true>(w_c1, fp.color.b[1], CW, c1, c2, wave, lane);
      __syncthreads();
      layer<CW, false>(w_c2, fp.color.b[2], 3, c2, c1, wave, lane);               // rgb pre-activation rows 0..2 (wave 0)
      __syncthreads();
      if (grp < 3 && valid) rgb[3 * ismp + grp] = sigmoidf(c1[grp * LDA + s]);
    }
    __syncthreads();
  }
}

template <int GEO, int NS, int SW>
static bool regw_shape_matches(const cn_field_params& p) {
  const cn_mlp &b = p.base, &s = p.semantics, &c = p.color;
  bool ok = p.grid.num_levels == 16 && p.geo_feat_dim == GEO && p.app_dim == 32 && b.num_layers == 2 &&
            b.dims[0] == 32 && b.dims[1] == 64 && b.dims[2] == 1 + GEO && s.num_layers == NS && s.dims[0] == GEO &&
            s.dims[NS] == 64 && c.num_layers == 3 && c.dims[0] == 16 + GEO + 32 && c.dims[1] == 64 &&
            c.dims[2] == 64 && c.dims[3] == 3;
  for (int l = 1; l < NS; ++l) ok = ok && s.dims[l] == SW;
  return ok;
}

constexpr size_t LDS_BYTES = ((size_t)(32 + 128 + 128 + 80 + 48) * LDA + 16 + 32) * sizeof(float);

}  // namespace rw
}  // namespace cn

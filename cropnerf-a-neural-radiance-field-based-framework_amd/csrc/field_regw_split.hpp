// FruitField evaluation with the MLP weights resident in registers, in the SPLIT-BF16 matrix arithmetic (included by
// field_simple.hip): cn_field_eval_mp(..., CN_MATRIX_SPLIT_BF16), the arithmetic FruitNerfModelConfig.matrix_precision
// defaults to for the renders that fill the device.  Every operand of a layer is a bf16 hi + a bf16 lo (16 mantissa
// bits), a . b ~ a_hi . b_hi + a_hi . b_lo + a_lo . b_hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- what
// render_split_kernel does for the default field shape (render_split.hpp), here for the shapes the fused renderers are not
// built for (fruit_nerf_method_big / _huge: geo 30, 3 x 128 semantic layers) and, for cross-checks, the default one.
//
// Same ownership as field_regw.hpp: wave w of a 512-thread workgroup owns row tile w % tiles of every layer and keeps those
// rows as A operands for the whole kernel -- K / 4 registers per layer again (a 32-wide K block is 4 registers of hi and
// 4 of lo), 168 for the big shape.  What changes is the activations: a 64-sample tile lives in LDS SAMPLE-major and
// ALREADY SPLIT, [sample][K + 8] bf16 twice (hi, lo), written by the wave that produced it (an accumulator lane holds
// four consecutive features of one sample: one 8-byte write each) -- so the B operand of a 32-wide K block is ONE
// ds_read_b128 of hi and one of lo instead of eight 4-byte reads, and an activation is split once, by its producer,
// instead of once per row tile that consumes it.  The fp32 form issues 8 matrix instructions of 32 cycles per 32-wide K
// block, this one 3 of 16.
//
// Internal feature order (the weight loaders permute columns to match; pad columns hold zero weights):
//   base output / semantic input  [logit | geo 0..GEO-1 | 0 ...] (32)   -- the logit column of the semantic layer is zero
//   colour input                  [logit | geo | 0 ... (32) | SH 16 | appearance 32 | 0 ...] (96)
// so the accumulator rows of the base MLP's second layer go to both buffers at the offset they already have.
// Density, rgb and the semantic logit leave from the accumulators (the semantic head is a 16 x 64 layer with one real row).
#pragma once

namespace cn {
namespace rws {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int TS = 64, NT = 512, NW = 8;

constexpr int pad16(int n) { return (n + 15) & ~15; }
constexpr int pad32(int n) { return (n + 31) & ~31; }
constexpr int stride_of(int kp) { return kp + 8; }  // bf16 elements per sample row: 16-byte aligned, conflict-free b128 reads

// column of the caller's weight matrix that internal input feature k multiplies, or -1 (zero weight)
enum ColMap { MAP_IDENT, MAP_SEM0, MAP_COL0 };
template <ColMap M, int GEO>
__device__ __forceinline__ int col_of(int k, int K) {
  if (M == MAP_IDENT) return k < K ? k : -1;
  if (M == MAP_SEM0) return (k >= 1 && k <= GEO) ? k - 1 : -1;
  // MAP_COL0: caller's order is [SH 16 | geo GEO | appearance 32]
  if (k >= 1 && k <= GEO) return 16 + k - 1;
  if (k >= 32 && k < 48) return k - 32;
  if (k >= 48 && k < 80) return 16 + GEO + (k - 48);
  return -1;
}

// A operands of one row tile: lane (i = lane & 15, q = lane >> 4) holds W[16 rt + i][32 kb + 8 q + e], e = 0..7, as hi and lo
template <int KP>
struct RowTile {
  bf16x8 hi[KP / 32], lo[KP / 32];
  template <ColMap M, int GEO>
  __device__ __forceinline__ void load(const float* __restrict__ W, int N, int K, int wave, int lane) {
    const int ntiles = pad16(N) / 16;
    const int i = lane & 15, q = lane >> 4, n = 16 * (wave % ntiles) + i;
#pragma unroll
    for (int kb = 0; kb < KP / 32; ++kb) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = col_of<M, GEO>(32 * kb + 8 * q + e, K);
        const float w = (n < N && c >= 0) ? W[(size_t)n * K + c] : 0.f;
        const __bf16 h = (__bf16)w;
        hi[kb][e] = h;
        lo[kb][e] = (__bf16)(w - (float)h);
      }
    }
  }
};

struct SplitBuf {
  __bf16* hi;
  __bf16* lo;
  int stride;  // bf16 elements per sample
};

// four consecutive features (k0 .. k0 + 3) of sample s, split and stored
__device__ __forceinline__ void store_split4(const SplitBuf& b, int s, int k0, const f32x4& v) {
  bf16x4 h, l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const __bf16 x = (__bf16)v[r];
    h[r] = x;
    l[r] = (__bf16)(v[r] - (float)x);
  }
  *reinterpret_cast<bf16x4*>(b.hi + s * b.stride + k0) = h;
  *reinterpret_cast<bf16x4*>(b.lo + s * b.stride + k0) = l;
}
__device__ __forceinline__ void store_split1(const SplitBuf& b, int s, int k, float v) {
  const __bf16 x = (__bf16)v;
  b.hi[s * b.stride + k] = x;
  b.lo[s * b.stride + k] = (__bf16)(v - (float)x);
}

// the column tiles (16 samples each) of this wave for a layer with `ntiles` row tiles (as field_regw.hpp)
__device__ __forceinline__ void col_tiles(int ntiles, int wave, int& ct_lo, int& ct_hi) {
  const int groups = NW / ntiles, grp = wave / ntiles;
  ct_lo = groups <= 4 ? grp * (4 / groups) : grp;
  ct_hi = groups <= 4 ? ct_lo + 4 / groups : (grp < 4 ? grp + 1 : grp);
}

template <int KP>
__device__ __forceinline__ f32x4 block(const RowTile<KP>& rt, const SplitBuf& in, int ct, f32x4 acc, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const __bf16* ph = in.hi + (16 * ct + i) * in.stride + 8 * q;
  const __bf16* pl = in.lo + (16 * ct + i) * in.stride + 8 * q;
#pragma unroll
  for (int kb = 0; kb < KP / 32; ++kb) {
    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(ph + 32 * kb);
    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pl + 32 * kb);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rt.hi[kb], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rt.hi[kb], bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rt.lo[kb], bh, acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ f32x4 bias4(const float* __restrict__ bias, int N, int rt0, int lane) {
  const int q = lane >> 4;
  f32x4 b;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = rt0 + 4 * q + r;
    b[r] = (bias && n < N) ? bias[n] : 0.f;
  }
  return b;
}

// out = act(b + W in), split, for this wave's (row tile, column tiles); rows >= N come out as zeros.  `out2` (optional)
// receives the same rows at the same offsets (the base MLP's output feeds two inputs).
template <int KP, bool RELU>
__device__ __forceinline__ void layer(const RowTile<KP>& rt, const float* __restrict__ bias, int N, const SplitBuf& in,
                                      const SplitBuf& out, const SplitBuf* out2, int wave, int lane) {
  const int ntiles = pad16(N) / 16, rt0 = 16 * (wave % ntiles);
  int ct_lo, ct_hi;
  col_tiles(ntiles, wave, ct_lo, ct_hi);
  const int i = lane & 15, q = lane >> 4;
  const f32x4 b4 = bias4(bias, N, rt0, lane);
#pragma unroll 1
  for (int ct = ct_lo; ct < ct_hi; ++ct) {
    f32x4 v = block<KP>(rt, in, ct, b4, lane);
    if (RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    store_split4(out, 16 * ct + i, rt0 + 4 * q, v);
    if (out2) store_split4(*out2, 16 * ct + i, rt0 + 4 * q, v);
  }
}

template <int GEO, int NS, int SW>
__global__ void __launch_bounds__(NT)
field_eval_regw_split_kernel(FieldDev fp, SceneDev sc, int app_mode, int sh_unit, const float* __restrict__ origins,
                             const float* __restrict__ directions, const int64_t* __restrict__ cam_idx,
                             const float* __restrict__ starts, const float* __restrict__ ends, long long num_rays, int S,
                             float* __restrict__ density, float* __restrict__ rgb, float* __restrict__ semantics,
                             float* __restrict__ positions) {
  constexpr int H = 64, HT = 64, CW = 64, APP = 32, CIN = 16 + GEO + APP, NG = 1 + GEO;
  constexpr int SWP = pad32(SW);
  static_assert(GEO <= 30 && SW <= 128 && SW % 32 == 0, "field shape");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // split-form buffers: [TS][K + 8] bf16, hi then lo
  auto carve = [](unsigned char*& p, int kp) {
    SplitBuf b;
    b.stride = stride_of(kp);
    b.hi = reinterpret_cast<__bf16*>(p);
    b.lo = b.hi + TS * b.stride;
    p += 2 * TS * b.stride * sizeof(__bf16);
    return b;
  };
  unsigned char* p = lds_raw;
  const SplitBuf ENC = carve(p, 32), X = carve(p, 128), Y = carve(p, 128), C = carve(p, 96), G = carve(p, 32);
  float* selv = reinterpret_cast<float*>(p);  // per-sample selector (0 / 1) of the tile
  float* scl = selv + TS;
  float* app_mean = scl + 16;
  const int lds_words = (int)((reinterpret_cast<unsigned char*>(app_mean + APP) - lds_raw) / 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = tid & 63, grp = wave;
  for (int e = tid; e < lds_words; e += NT) reinterpret_cast<unsigned*>(lds_raw)[e] = 0u;
  __syncthreads();
  if (tid < CN_MAX_LEVELS) scl[tid] = fp.grid.scale[tid];
  if (app_mode == CN_APP_MEAN && tid < APP) {
    float m = 0.f;
    for (int n = 0; n < fp.num_images; ++n) m += fp.appearance[(long long)n * APP + tid];
    app_mean[tid] = m / (float)fp.num_images;
  }
  // ---- the weights of this wave's row tiles, split, for the whole kernel ----------------------------------------------------
  RowTile<32> w_b0;
  RowTile<64> w_b1;
  RowTile<32> w_s0;
  RowTile<SWP> w_s1;
  RowTile<SWP> w_s2;  // third semantic layer (NS == 3)
  RowTile<64> w_sh;   // semantic head: one real row
  RowTile<96> w_c0;
  RowTile<64> w_c1;
  RowTile<64> w_c2;
  w_b0.template load<MAP_IDENT, GEO>(fp.base.w[0], H, 32, wave, lane);
  w_b1.template load<MAP_IDENT, GEO>(fp.base.w[1], NG, H, wave, lane);
  w_s0.template load<MAP_SEM0, GEO>(fp.sem.w[0], SW, GEO, wave, lane);
  w_s1.template load<MAP_IDENT, GEO>(fp.sem.w[1], NS == 3 ? SW : HT, SW, wave, lane);
  if (NS == 3) w_s2.template load<MAP_IDENT, GEO>(fp.sem.w[2], HT, SW, wave, lane);
  w_sh.template load<MAP_IDENT, GEO>(fp.sem_head_w, 1, HT, wave, lane);
  w_c0.template load<MAP_COL0, GEO>(fp.color.w[0], CW, CIN, wave, lane);
  w_c1.template load<MAP_IDENT, GEO>(fp.color.w[1], CW, CW, wave, lane);
  w_c2.template load<MAP_IDENT, GEO>(fp.color.w[2], 3, CW, wave, lane);
  __syncthreads();

  const long long total = num_rays * (long long)S;
  const long long ntiles = (total + TS - 1) / TS;
  const int i = lane & 15, q = lane >> 4;
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
    if (grp == 0) selv[s] = sel ? 1.f : 0.f;
    // ---- inputs: two grid levels per thread; SH and appearance columns of the colour input -----------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int l = grp + NW * h;
      const float2 f = hash_level_any(fp.grid, l, px, py, pz);
      store_split1(ENC, s, 2 * l, f.x);
      store_split1(ENC, s, 2 * l + 1, f.y);
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
        for (int k = 0; k < 16; k += 4) store_split4(C, s, 32 + k, f32x4{sh[k], sh[k + 1], sh[k + 2], sh[k + 3]});
      }
      const float* emb = app_mode == CN_APP_PER_CAMERA ? fp.appearance + cam_idx[r] * (long long)APP : nullptr;
      {  // four appearance columns per wave
        f32x4 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 4 * grp + e;
          a[e] = app_mode == CN_APP_MEAN ? app_mean[k] : (emb ? emb[k] : 0.f);
        }
        store_split4(C, s, 48 + 4 * grp, a);
      }
    }
    __syncthreads();
    layer<32, true>(w_b0, fp.base.b[0], H, ENC, X, nullptr, wave, lane);  // h1 -> X
    __syncthreads();
    {  // logit | geo -> G and the colour input; the density leaves from the accumulator (row 0: row tile 0, q == 0, r == 0)
      constexpr int ntl = pad16(NG) / 16;
      const int rt0 = 16 * (wave % ntl);
      int ct_lo, ct_hi;
      col_tiles(ntl, wave, ct_lo, ct_hi);
      const f32x4 b4 = bias4(fp.base.b[1], NG, rt0, lane);
#pragma unroll 1
      for (int ct = ct_lo; ct < ct_hi; ++ct) {
        const f32x4 v = block<64>(w_b1, X, ct, b4, lane);
        const int sc_ = 16 * ct + i;
        store_split4(G, sc_, rt0 + 4 * q, v);
        if (rgb) store_split4(C, sc_, rt0 + 4 * q, v);
        if (density && rt0 == 0 && q == 0 && tile * TS + sc_ < total) density[tile * TS + sc_] = expf(v[0]) * selv[sc_];
      }
    }
    __syncthreads();
    layer<32, true>(w_s0, fp.sem.b[0], SW, G, X, nullptr, wave, lane);  // s1 -> X
    __syncthreads();
    if (NS == 3) {
      layer<SWP, true>(w_s1, fp.sem.b[1], SW, X, Y, nullptr, wave, lane);   // s2 -> Y
      __syncthreads();
      layer<SWP, false>(w_s2, fp.sem.b[2], HT, Y, X, nullptr, wave, lane);  // s3 -> X
    } else {
      layer<SWP, false>(w_s1, fp.sem.b[1], HT, X, Y, nullptr, wave, lane);  // s2 -> Y
    }
    __syncthreads();
    const SplitBuf& sout = NS == 3 ? X : Y;
    if (semantics && wave < 4) {  // head: one row tile, wave w takes column tile w; row 0 sits in q == 0, r == 0
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (q == 0) b4[0] = fp.sem_head_b[0];
      const f32x4 v = block<64>(w_sh, sout, wave, b4, lane);
      const long long o = tile * TS + 16 * wave + i;
      if (q == 0 && o < total) semantics[o] = v[0];
    }
    if (rgb) {
      const SplitBuf& c1 = NS == 3 ? Y : X;  // the buffer the semantic output is NOT in
      const SplitBuf& c2 = NS == 3 ? X : Y;
      layer<96, true>(w_c0, fp.color.b[0], CW, C, c1, nullptr, wave, lane);
      __syncthreads();  // (also: the head above has read `sout` == c2 in every wave)
      layer<64, true>(w_c1, fp.color.b[1], CW, c1, c2, nullptr, wave, lane);
      __syncthreads();
      if (wave < 4) {  // rgb rows 0..2 of one row tile
        const f32x4 v = block<64>(w_c2, c2, wave, bias4(fp.color.b[2], 3, 0, lane), lane);
        const long long o = tile * TS + 16 * wave + i;
        if (q == 0 && o < total) {
          rgb[3 * o + 0] = sigmoidf(v[0]);
          rgb[3 * o + 1] = sigmoidf(v[1]);
          rgb[3 * o + 2] = sigmoidf(v[2]);
        }
      }
    }
    __syncthreads();
  }
}

constexpr size_t LDS_BYTES = (size_t)2 * TS * (stride_of(32) + 128 + 8 + 128 + 8 + stride_of(96) + stride_of(32)) * 2 +
                             (TS + 16 + 32) * sizeof(float);

}  // namespace rws
}  // namespace cn

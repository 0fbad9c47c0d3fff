// cn_field_eval / cn_proposal_density: the straightforward evaluation of FruitField / HashMLPDensityField,
// one lane per sample, fp32 VALU, activations staged in LDS as [feature][lane] (conflict-free), weights read through
// wave-uniform (scalar) loads.  Handles every layer shape the reference configs can produce
// (fruit_nerf/fruit_nerf_config.py:29-172).  This is the general, obviously-correct path; the throughput path is
// render_fused.hip, which is checked against both the CPU oracle and this kernel.
//
// Reference: fruit_nerf/fruit_field.py:169-302 (FruitField), fruit_nerf/fruit_nerf.py:118-142 (proposal nets).
#include <cstdlib>
#include <cstring>

#include <mutex>

#include "cn_common.hpp"
#include "wave_ops.hpp"

namespace cn {

constexpr int KMAX = 176;   // widest activation vector (16 + geo 30 + app 128 = 174)
constexpr int GMAX = 32;    // 1 + geo_feat_dim

struct MlpDev {
  int num_layers;
  int dims[CN_MAX_LAYERS + 1];
  const float* w[CN_MAX_LAYERS];
  const float* b[CN_MAX_LAYERS];
};

inline MlpDev make_mlp_dev(const cn_mlp& m) {
  MlpDev d;
  d.num_layers = m.num_layers;
  for (int i = 0; i <= CN_MAX_LAYERS; ++i) d.dims[i] = m.dims[i];
  for (int i = 0; i < CN_MAX_LAYERS; ++i) {
    d.w[i] = m.weight[i];
    d.b[i] = m.bias[i];
  }
  return d;
}

// y[n][lane] = act(b[n] + sum_k W[n][k] x[k][lane]);  x, y: LDS [.][64]
__device__ __forceinline__ void dense_layer(const float* __restrict__ W, const float* __restrict__ b, int K, int N,
                                            const float* x, float* y, bool relu, int lane) {
  for (int n0 = 0; n0 < N; n0 += 8) {
    float acc[8];
    int nn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      nn[j] = min(n0 + j, N - 1);
      acc[j] = b[nn[j]];
    }
    for (int k = 0; k < K; ++k) {
      float xk = x[k * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(W[nn[j] * K + k], xk, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (n0 + j < N) y[(n0 + j) * 64 + lane] = relu ? fmaxf(acc[j], 0.f) : acc[j];
    }
  }
}

// runs all layers of `m`; input in `in` (LDS), ping-pongs between a and b; returns pointer to the output rows
__device__ __forceinline__ float* run_mlp(const MlpDev& m, const float* in, float* a, float* b, int lane) {
  const float* x = in;
  float* y = a;
  for (int l = 0; l < m.num_layers; ++l) {
    y = (l & 1) ? b : a;
    dense_layer(m.w[l], m.b[l], m.dims[l], m.dims[l + 1], x, y, l < m.num_layers - 1, lane);
    x = y;
  }
  return y;
}

__device__ __forceinline__ void encode_grid(const GridDev& g, float px, float py, float pz, float* out, int lane) {
  for (int l = 0; l < g.num_levels; ++l) {
    float2 f = hash_level_any(g, l, px, py, pz);
    out[(2 * l) * 64 + lane] = f.x;
    out[(2 * l + 1) * 64 + lane] = f.y;
  }
}

struct FieldDev {
  GridDev grid;
  MlpDev base, sem, color;
  const float* sem_head_w;
  const float* sem_head_b;
  const float* appearance;
  int num_images, app_dim, geo;
};

__global__ void __launch_bounds__(64)
field_eval_kernel(FieldDev fp, SceneDev sc, int app_mode, int sh_unit, const float* __restrict__ origins,
                  const float* __restrict__ directions, const int64_t* __restrict__ cam_idx,
                  const float* __restrict__ starts, const float* __restrict__ ends, long long num_rays, int S,
                  float* __restrict__ density, float* __restrict__ rgb, float* __restrict__ semantics,
                  float* __restrict__ positions) {
  extern __shared__ __align__(16) float lds[];
  float* bufG = lds;                   // [GMAX][64] base output (density logit + geo)
  float* bufA = bufG + GMAX * 64;      // [KMAX][64]
  float* bufB = bufA + KMAX * 64;      // [KMAX][64]
  float* app_mean = bufB + KMAX * 64;  // [app_dim]
  const int lane = threadIdx.x;
  if (app_mode == CN_APP_MEAN) {
    for (int j = lane; j < fp.app_dim; j += 64) {
      float s = 0.f;
      for (int i = 0; i < fp.num_images; ++i) s += fp.appearance[(long long)i * fp.app_dim + j];
      app_mean[j] = s / (float)fp.num_images;
    }
  }
  __syncthreads();
  const long long total = num_rays * (long long)S;
  const long long nblk = (total + 63) / 64;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    long long i = blk * 64 + lane;
    bool valid = i < total;
    long long ic = valid ? i : total - 1;
    long long r = ic / S;
    float ox = origins[3 * r], oy = origins[3 * r + 1], oz = origins[3 * r + 2];
    float dx = directions[3 * r], dy = directions[3 * r + 1], dz = directions[3 * r + 2];
    float mid = (starts[ic] + ends[ic]) / 2.f;
    float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
    if (positions && valid) {
      positions[3 * i + 0] = px;
      positions[3 * i + 1] = py;
      positions[3 * i + 2] = pz;
    }
    bool sel = normalize_position(sc, px, py, pz);
    encode_grid(fp.grid, px, py, pz, bufA, lane);
    // base MLP: input bufA, layers alternate bufB / bufG so that the final output lands in bufG
    {
      const float* x = bufA;
      for (int l = 0; l < fp.base.num_layers; ++l) {
        bool last = l == fp.base.num_layers - 1;
        float* y = last ? bufG : ((x == bufA) ? bufB : bufA);
        dense_layer(fp.base.w[l], fp.base.b[l], fp.base.dims[l], fp.base.dims[l + 1], x, y, !last, lane);
        x = y;
      }
    }
    float den = expf(bufG[lane]) * (sel ? 1.f : 0.f);
    if (density && valid) density[i] = den;
    // semantics: mlp_semantics(geo) -> Linear(Ht,1)
    {
      float* out = run_mlp(fp.sem, bufG + 64, bufA, bufB, lane);
      int ht = fp.sem.dims[fp.sem.num_layers];
      float s = fp.sem_head_b[0];
      for (int k = 0; k < ht; ++k) s = fmaf(fp.sem_head_w[k], out[k * 64 + lane], s);
      if (semantics && valid) semantics[i] = s;
    }
    // colour: [SH16 | geo | appearance] -> mlp_head -> sigmoid
    if (rgb) {
      float sx = dx, sy = dy, sz = dz;
      if (!sh_unit) {
        sx = (dx + 1.f) / 2.f;
        sy = (dy + 1.f) / 2.f;
        sz = (dz + 1.f) / 2.f;
      }
      float sh[16];
      sh_deg4(sx, sy, sz, sh);
#pragma unroll
      for (int k = 0; k < 16; ++k) bufA[k * 64 + lane] = sh[k];
      for (int k = 0; k < fp.geo; ++k) bufA[(16 + k) * 64 + lane] = bufG[(1 + k) * 64 + lane];
      const float* emb = nullptr;
      if (app_mode == CN_APP_PER_CAMERA) emb = fp.appearance + cam_idx[r] * (long long)fp.app_dim;
      for (int k = 0; k < fp.app_dim; ++k) {
        float a = app_mode == CN_APP_MEAN ? app_mean[k] : (emb ? emb[k] : 0.f);
        bufA[(16 + fp.geo + k) * 64 + lane] = a;
      }
      // layers: bufA -> bufB -> bufA -> ...
      const float* x = bufA;
      float* y = bufB;
      for (int l = 0; l < fp.color.num_layers; ++l) {
        y = (x == bufA) ? bufB : bufA;
        dense_layer(fp.color.w[l], fp.color.b[l], fp.color.dims[l], fp.color.dims[l + 1], x, y,
                    l < fp.color.num_layers - 1, lane);
        x = y;
      }
      if (valid) {
        rgb[3 * i + 0] = sigmoidf(y[0 * 64 + lane]);
        rgb[3 * i + 1] = sigmoidf(y[1 * 64 + lane]);
        rgb[3 * i + 2] = sigmoidf(y[2 * 64 + lane]);
      }
    }
  }
}

struct DensityDev {
  GridDev grid;
  MlpDev mlp;
};

__global__ void __launch_bounds__(64)
proposal_density_kernel(DensityDev dp, SceneDev sc, const float* __restrict__ origins,
                        const float* __restrict__ directions, const float* __restrict__ starts,
                        const float* __restrict__ ends, long long num_rays, int S, float* __restrict__ density) {
  __shared__ float bufA[64 * 64];
  __shared__ float bufB[64 * 64];
  const int lane = threadIdx.x;
  const long long total = num_rays * (long long)S;
  const long long nblk = (total + 63) / 64;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    long long i = blk * 64 + lane;
    bool valid = i < total;
    long long ic = valid ? i : total - 1;
    long long r = ic / S;
    float mid = (starts[ic] + ends[ic]) / 2.f;
    float px = origins[3 * r] + directions[3 * r] * mid;
    float py = origins[3 * r + 1] + directions[3 * r + 1] * mid;
    float pz = origins[3 * r + 2] + directions[3 * r + 2] * mid;
    bool sel = normalize_position(sc, px, py, pz);
    encode_grid(dp.grid, px, py, pz, bufA, lane);
    const float* x = bufA;
    float* y = bufB;
    for (int l = 0; l < dp.mlp.num_layers; ++l) {
      y = (x == bufA) ? bufB : bufA;
      dense_layer(dp.mlp.w[l], dp.mlp.b[l], dp.mlp.dims[l], dp.mlp.dims[l + 1], x, y, l < dp.mlp.num_layers - 1, lane);
      x = y;
    }
    if (valid) density[i] = expf(y[lane]) * (sel ? 1.f : 0.f);
  }
}

static int validate_mlp(const cn_mlp& m, const char* name, int in_dim, int out_dim, int max_width) {
  CN_REQUIRE(m.num_layers >= 1 && m.num_layers <= CN_MAX_LAYERS, CN_ERR_INVALID, "%s: num_layers %d", name,
             m.num_layers);
  CN_REQUIRE(m.dims[0] == in_dim, CN_ERR_INVALID, "%s: in_dim %d, expected %d", name, m.dims[0], in_dim);
  if (out_dim > 0)
    CN_REQUIRE(m.dims[m.num_layers] == out_dim, CN_ERR_INVALID, "%s: out_dim %d, expected %d", name,
               m.dims[m.num_layers], out_dim);
  for (int i = 0; i <= m.num_layers; ++i)
    CN_REQUIRE(m.dims[i] >= 1 && m.dims[i] <= max_width, CN_ERR_UNSUPPORTED, "%s: layer width %d > %d", name, m.dims[i],
               max_width);
  for (int i = 0; i < m.num_layers; ++i)
    CN_REQUIRE(m.weight[i] && m.bias[i], CN_ERR_INVALID, "%s: null weight/bias in layer %d", name, i);
  return CN_OK;
}

int validate_grid(const cn_grid& g, const char* name) { return check_grid(g, false, name); }


// ------------------------------------------------------------------------------------------------------------------------
// Shape-generic FruitField on the fp32 matrix cores (the path fruit_nerf_method_big / _huge take).
// One 512-thread workgroup walks 64-sample tiles.  Activations live in LDS as [feature][68] (64 samples + 4 pad), three
// 128-row buffers + one for the base output; every dense layer is staged 64 output rows at a time into LDS as
// [row][Kpad + 4] (zero-padded to multiples of 16) and evaluated with v_mfma_f32_16x16x4_f32: wave w owns the 16-sample
// column tile w and walks the row tiles.  Pad rows come out as exact zeros, so they are valid pad inputs of the next
// layer.  Works for any layer widths <= 128 (the reference's configs: 64 / 128); wider shapes fall back to
// field_eval_kernel.
// ------------------------------------------------------------------------------------------------------------------------
namespace gm {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TS = 64, LDA = 68, NT = 512, NW = NT / 64, WMAX = 128, WS_MAX = WMAX + 4;
constexpr int ROWS_G = 48;
constexpr size_t LDS_FLOATS = (size_t)3 * WMAX * LDA + ROWS_G * LDA + 64 * WS_MAX + WMAX;

// (Prefetching the next pass's weights into registers during the MFMAs was tried: no gain -- the cost of a pass is the
// L2 -> LDS volume of re-staging 185 KB of weights per 64-sample tile, not its latency.)
__device__ __forceinline__ void dense_mfma(const float* __restrict__ Wg, const float* __restrict__ bg_, int K_, int N_,
                                           const float* in, float* out, bool relu, float* wbuf, int tid,
                                           int debug_skip = 0) {
  const float* __restrict__ bg = bg_;
  const int K = K_, N = N_;
  const int Kp = (K + 15) & ~15, ws = Kp + 4;
  const int lane = tid & 63, wave = tid >> 6;
  // 8 waves (2 per SIMD): column tile = wave & 3, the two waves of a column tile split the row tiles
  const int i = lane & 15, q = lane >> 4, s0 = 16 * (wave & 3), nt0 = wave >> 2;
  for (int p0 = 0; p0 < N; p0 += 64) {
    const int rows = min(64, N - p0), rows_p = (rows + 15) & ~15;
    __syncthreads();  // the previous pass / layer has finished with wbuf, and `in` is complete
    for (int row = wave; row < rows_p; row += NW)      // wave per row, lanes along k: coalesced, no division
      for (int k = lane; k < Kp; k += 64)
        wbuf[row * ws + k] = (row < rows && k < K) ? Wg[(size_t)(p0 + row) * K + k] : 0.f;
    __syncthreads();
    for (int nt = nt0; nt < rows_p / 16 && !(debug_skip & 32); nt += 2) {
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = p0 + 16 * nt + 4 * q + r;
        acc[r] = n < N ? bg[n] : 0.f;
      }
      // operands of round kb + 1 are fetched from LDS while the four MFMAs of round kb run
      f32x4 a = *reinterpret_cast<const f32x4*>(wbuf + (16 * nt + i) * ws + 4 * q);
      float b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = in[(4 * q + e) * LDA + s0 + i];
      for (int kb = 0; kb < Kp / 16; ++kb) {
        const int kn = kb + 1 < Kp / 16 ? kb + 1 : kb;
        const f32x4 an = *reinterpret_cast<const f32x4*>(wbuf + (16 * nt + i) * ws + 16 * kn + 4 * q);
        float bn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bn[e] = in[(16 * kn + 4 * q + e) * LDA + s0 + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = a[e];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[e], acc, 0, 0, 0);
        }
        a = an;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = bn[e];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r];
        if (relu) v = fmaxf(v, 0.f);
        out[(p0 + 16 * nt + 4 * q + r) * LDA + s0 + i] = v;
      }
    }
  }
}

// (A variant of dense_mfma that read the weights straight from global memory as MFMA A operands -- one barrier per layer
// instead of two per 64-row pass, but four waves fetch each weight row -- measured 48 vs 29.5 ms per 65k big-method rays and
// was removed.)

__global__ void __launch_bounds__(NT)
field_eval_mfma_kernel(FieldDev fp, SceneDev sc, int app_mode, int sh_unit, const float* __restrict__ origins,
                       const float* __restrict__ directions, const int64_t* __restrict__ cam_idx,
                       const float* __restrict__ starts, const float* __restrict__ ends, long long num_rays, int S,
                       float* __restrict__ density, float* __restrict__ rgb, float* __restrict__ semantics,
                       float* __restrict__ positions, int debug_skip) {
  extern __shared__ __align__(16) float lds[];
  float* bufA = lds;
  float* bufB = bufA + WMAX * LDA;
  float* bufC = bufB + WMAX * LDA;
  float* bufG = bufC + WMAX * LDA;        // base output: row 0 = density logit, rows 1..geo = geo features, rest zero
  float* wbuf = bufG + ROWS_G * LDA;
  float* app_mean = wbuf + 64 * WS_MAX;
  const int tid = threadIdx.x, s = tid & 63, grp = tid >> 6;
  for (int e = tid; e < (int)(3 * WMAX * LDA + ROWS_G * LDA); e += NT) lds[e] = 0.f;
  if (app_mode == CN_APP_MEAN) {
    for (int j = tid; j < fp.app_dim; j += NT) {
      float m = 0.f;
      for (int n = 0; n < fp.num_images; ++n) m += fp.appearance[(long long)n * fp.app_dim + j];
      app_mean[j] = m / (float)fp.num_images;
    }
  }
  __syncthreads();
  const int enc_dim = 2 * fp.grid.num_levels;
  const int cin_dim = 16 + fp.geo + fp.app_dim;
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
    __syncthreads();  // the previous tile's readers of bufA / bufC are done
    for (int l = grp; l < fp.grid.num_levels && !(debug_skip & 64); l += NW) {
      const float2 f = hash_level_any(fp.grid, l, px, py, pz);
      bufA[(2 * l) * LDA + s] = f.x;
      bufA[(2 * l + 1) * LDA + s] = f.y;
    }
    for (int k = enc_dim + grp; k < ((enc_dim + 15) & ~15); k += NW) bufA[k * LDA + s] = 0.f;
    // colour input rows that do not depend on the base MLP: SH (rows 0..15) and appearance
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
      const float* emb = app_mode == CN_APP_PER_CAMERA ? fp.appearance + cam_idx[r] * (long long)fp.app_dim : nullptr;
      for (int k = grp; k < fp.app_dim; k += NW)
        bufC[(16 + fp.geo + k) * LDA + s] = app_mode == CN_APP_MEAN ? app_mean[k] : (emb ? emb[k] : 0.f);
      for (int k = cin_dim + grp; k < ((cin_dim + 15) & ~15); k += NW) bufC[k * LDA + s] = 0.f;
    }
    // ---- base MLP: bufA -> (bufB -> bufA ...) -> bufG ------------------------------------------------------------------
    {
      const float* x = bufA;
      for (int l = 0; l < fp.base.num_layers; ++l) {
        const bool last = l == fp.base.num_layers - 1;
        float* y = last ? bufG : (x == bufA ? bufB : bufA);
        dense_mfma(fp.base.w[l], fp.base.b[l], fp.base.dims[l], fp.base.dims[l + 1], x, y, !last, wbuf, tid, debug_skip);
        x = y;
      }
    }
    __syncthreads();
    if (grp == 0 && density && valid) density[ismp] = expf(bufG[s]) * (sel ? 1.f : 0.f);
    if (rgb)
      for (int k = grp; k < fp.geo; k += NW) bufC[(16 + k) * LDA + s] = bufG[(1 + k) * LDA + s];
    // ---- semantics: mlp_semantics(geo) -> Linear(Ht, 1) --------------------------------------------------------------------
    {
      const float* x = bufG + LDA;
      for (int l = 0; l < fp.sem.num_layers; ++l) {
        float* y = (x == bufA) ? bufB : bufA;
        dense_mfma(fp.sem.w[l], fp.sem.b[l], fp.sem.dims[l], fp.sem.dims[l + 1], x, y, l < fp.sem.num_layers - 1, wbuf, tid,
                   debug_skip);
        x = y;
      }
      __syncthreads();
      if (grp == 0 && semantics && valid) {
        const int ht = fp.sem.dims[fp.sem.num_layers];
        float v = fp.sem_head_b[0];
        for (int k = 0; k < ht; ++k) v = fmaf(fp.sem_head_w[k], x[k * LDA + s], v);
        semantics[ismp] = v;
      }
    }
    // ---- colour: [SH16 | geo | appearance] -> mlp_head -> sigmoid -----------------------------------------------------------
    if (rgb) {
      const float* x = bufC;
      for (int l = 0; l < fp.color.num_layers; ++l) {
        float* y = (x == bufA) ? bufB : bufA;
        dense_mfma(fp.color.w[l], fp.color.b[l], fp.color.dims[l], fp.color.dims[l + 1], x, y,
                   l < fp.color.num_layers - 1, wbuf, tid, debug_skip);
        x = y;
      }
      __syncthreads();
      if (grp < 3 && valid) rgb[3 * ismp + grp] = sigmoidf(x[grp * LDA + s]);
    }
  }
}

// every layer's input and output width fits the 128-row LDS buffers
static bool mfma_generic_ok(const cn_field_params& p) {
  auto ok = [](const cn_mlp& m) {
    for (int l = 0; l <= m.num_layers; ++l)
      if (m.dims[l] > WMAX) return false;
    return true;
  };
  return ok(p.base) && ok(p.semantics) && ok(p.color) && 1 + p.geo_feat_dim <= 32 && 2 * p.grid.num_levels <= WMAX &&
         16 + p.geo_feat_dim + p.app_dim <= WMAX && p.app_dim <= WMAX;
}
}  // namespace gm

}  // namespace cn
#include "field_regw.hpp"
#include "field_regw_split.hpp"
namespace cn {

int validate_field(const cn_field_params& p) {
  int rc = validate_grid(p.grid, "field grid");
  if (rc) return rc;
  CN_REQUIRE(p.geo_feat_dim >= 1 && p.geo_feat_dim < GMAX, CN_ERR_UNSUPPORTED, "geo_feat_dim %d", p.geo_feat_dim);
  CN_REQUIRE(p.app_dim >= 0 && 16 + p.geo_feat_dim + p.app_dim <= KMAX, CN_ERR_UNSUPPORTED, "app_dim %d", p.app_dim);
  if ((rc = validate_mlp(p.base, "mlp_base_mlp", 2 * p.grid.num_levels, 1 + p.geo_feat_dim, KMAX))) return rc;
  if ((rc = validate_mlp(p.semantics, "mlp_semantics", p.geo_feat_dim, 0, KMAX))) return rc;
  if ((rc = validate_mlp(p.color, "mlp_head", 16 + p.geo_feat_dim + p.app_dim, 3, KMAX))) return rc;
  CN_REQUIRE(p.sem_head_weight && p.sem_head_bias, CN_ERR_INVALID, "null semantic head");
  CN_REQUIRE(p.app_dim == 0 || (p.appearance && p.num_images > 0), CN_ERR_INVALID, "null appearance embedding");
  return CN_OK;
}

FieldDev make_field_dev(const cn_field_params& p) {
  FieldDev f;
  f.grid = make_grid_dev(p.grid);
  f.base = make_mlp_dev(p.base);
  f.sem = make_mlp_dev(p.semantics);
  f.color = make_mlp_dev(p.color);
  f.sem_head_w = p.sem_head_weight;
  f.sem_head_b = p.sem_head_bias;
  f.appearance = p.appearance;
  f.num_images = p.num_images;
  f.app_dim = p.app_dim;
  f.geo = p.geo_feat_dim;
  return f;
}

}  // namespace cn

extern "C" int cn_field_eval(const cn_field_params* params, const cn_scene* scene, int32_t app_mode,
                             int32_t sh_unit_dir, const float* origins, const float* directions,
                             const int64_t* camera_indices, const float* starts, const float* ends, int64_t num_rays,
                             int32_t num_samples, float* density, float* rgb, float* semantics, float* positions,
                             cn_stream_t stream) {
  return cn_field_eval_mp(params, scene, app_mode, sh_unit_dir, origins, directions, camera_indices, starts, ends, num_rays,
                          num_samples, density, rgb, semantics, positions, CN_MATRIX_FP32, stream);
}

extern "C" int cn_field_eval_mp(const cn_field_params* params, const cn_scene* scene, int32_t app_mode,
                                int32_t sh_unit_dir, const float* origins, const float* directions,
                                const int64_t* camera_indices, const float* starts, const float* ends, int64_t num_rays,
                                int32_t num_samples, float* density, float* rgb, float* semantics, float* positions,
                                int32_t matrix_precision, cn_stream_t stream) {
  CN_REQUIRE(params && scene && origins && directions && starts && ends, CN_ERR_INVALID, "cn_field_eval: null input");
  CN_REQUIRE(matrix_precision == CN_MATRIX_FP32 || matrix_precision == CN_MATRIX_SPLIT_BF16 || matrix_precision == CN_MATRIX_F16,
             CN_ERR_INVALID, "cn_field_eval: matrix_precision %d", matrix_precision);
  CN_REQUIRE(num_samples > 0, CN_ERR_INVALID, "cn_field_eval: num_samples must be > 0");
  CN_REQUIRE(app_mode >= CN_APP_ZEROS && app_mode <= CN_APP_PER_CAMERA, CN_ERR_INVALID, "cn_field_eval: app_mode %d",
             app_mode);
  // fruit_field.py:241-242 raises AttributeError when camera indices are missing
  CN_REQUIRE(app_mode != CN_APP_PER_CAMERA || camera_indices, CN_ERR_INVALID, "Camera indices are not provided.");
  int rc = cn::validate_field(*params);
  if (rc) return rc;
  if (num_rays <= 0) return CN_OK;
  size_t lds = (size_t)(cn::GMAX + 2 * cn::KMAX) * 64 * sizeof(float) + 256 * sizeof(float);
  const size_t lds_mfma = cn::gm::LDS_FLOATS * sizeof(float);
  // one-time kernel attributes, per device (the entry points are re-entrant)
  static cn::PerDevice<int> attrs;
  rc = attrs.get(
      [&](int, int&) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::field_eval_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::gm::field_eval_mfma_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::rw::field_eval_regw_kernel<15, 2, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cn::rw::LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::rw::field_eval_regw_kernel<30, 3, 128>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cn::rw::LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(cn::rws::field_eval_regw_split_kernel<15, 2, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cn::rws::LDS_BYTES);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(cn::rws::field_eval_regw_split_kernel<30, 3, 128>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)cn::rws::LDS_BYTES);
      },
      nullptr, "cn_field_eval");
  if (rc) return rc;
  // Implementations, fastest first (CN_FIELD_EVAL_IMPL = regw | mfma | scalar forces one; the tests compare them):
  //   regw   weights resident in registers, the two field shapes of the reference's configs (field_regw.hpp)
  //   mfma   any widths <= 128, weights staged through LDS per tile
  //   scalar any widths <= 176, scalar FMAs
  const char* impl = getenv("CN_FIELD_EVAL_IMPL");
  const bool want = impl != nullptr;
  if (!want || std::strcmp(impl, "regw") == 0) {
    const bool def = cn::rw::regw_shape_matches<15, 2, 64>(*params), big = cn::rw::regw_shape_matches<30, 3, 128>(*params);
    // The 16-bit matrix modes (the model's default for the renders that fill the device is split-bf16): the two shapes of the
    // reference's configs in field_regw_split.hpp.  CN_MATRIX_F16 has no shape-generic kernel of its own and takes the same
    // one (bf16 hi + lo operands: more precise than fp16 operands, and the faster of the two); other shapes stay exact fp32.
    if ((def || big) && matrix_precision != CN_MATRIX_FP32) {
      const long long ntiles = (num_rays * (long long)num_samples + cn::rws::TS - 1) / cn::rws::TS;
      const dim3 grid(cn::grid_for(ntiles, 1, 256)), block(cn::rws::NT);
      if (def)
        hipLaunchKernelGGL((cn::rws::field_eval_regw_split_kernel<15, 2, 64>), grid, block, cn::rws::LDS_BYTES,
                           cn::as_stream(stream), cn::make_field_dev(*params), cn::make_scene_dev(*scene), app_mode,
                           sh_unit_dir, origins, directions, camera_indices, starts, ends, (long long)num_rays,
                           num_samples, density, rgb, semantics, positions);
      else
        hipLaunchKernelGGL((cn::rws::field_eval_regw_split_kernel<30, 3, 128>), grid, block, cn::rws::LDS_BYTES,
                           cn::as_stream(stream), cn::make_field_dev(*params), cn::make_scene_dev(*scene), app_mode,
                           sh_unit_dir, origins, directions, camera_indices, starts, ends, (long long)num_rays,
                           num_samples, density, rgb, semantics, positions);
      return cn::check_launch("cn_field_eval_mp");
    }
    if (def || big) {
      const long long ntiles = (num_rays * (long long)num_samples + cn::rw::TS - 1) / cn::rw::TS;
      const dim3 grid(cn::grid_for(ntiles, 1, 256)), block(cn::rw::NT);
      if (def)
        hipLaunchKernelGGL((cn::rw::field_eval_regw_kernel<15, 2, 64>), grid, block, cn::rw::LDS_BYTES,
                           cn::as_stream(stream), cn::make_field_dev(*params), cn::make_scene_dev(*scene), app_mode,
                           sh_unit_dir, origins, directions, camera_indices, starts, ends, (long long)num_rays,
                           num_samples, density, rgb, semantics, positions);
      else
        hipLaunchKernelGGL((cn::rw::field_eval_regw_kernel<30, 3, 128>), grid, block, cn::rw::LDS_BYTES,
                           cn::as_stream(stream), cn::make_field_dev(*params), cn::make_scene_dev(*scene), app_mode,
                           sh_unit_dir, origins, directions, camera_indices, starts, ends, (long long)num_rays,
                           num_samples, density, rgb, semantics, positions);
      return cn::check_launch("cn_field_eval");
    }
  }
  if (cn::gm::mfma_generic_ok(*params) && !(impl && std::strcmp(impl, "scalar") == 0)) {
    long long ntiles = (num_rays * (long long)num_samples + cn::gm::TS - 1) / cn::gm::TS;
    hipLaunchKernelGGL(cn::gm::field_eval_mfma_kernel, dim3(cn::grid_for(ntiles, 1, 256)), dim3(cn::gm::NT), lds_mfma,
                       cn::as_stream(stream), cn::make_field_dev(*params), cn::make_scene_dev(*scene), app_mode,
                       sh_unit_dir, origins, directions, camera_indices, starts, ends, (long long)num_rays, num_samples,
                       density, rgb, semantics, positions, [] {
                         const char* e = getenv("CN_DEBUG_SKIP");  // profiling aid: 32 no MFMA loops, 64 no gathers
                         return e ? atoi(e) : 0;
                       }());
    return cn::check_launch("cn_field_eval");
  }
  long long nblk = (num_rays * (long long)num_samples + 63) / 64;
  hipLaunchKernelGGL(cn::field_eval_kernel, dim3(cn::grid_for(nblk, 1, 256 * 8)), dim3(64), lds, cn::as_stream(stream),
                     cn::make_field_dev(*params), cn::make_scene_dev(*scene), app_mode, sh_unit_dir, origins,
                     directions, camera_indices, starts, ends, (long long)num_rays, num_samples, density, rgb,
                     semantics, positions);
  return cn::check_launch("cn_field_eval");
}

extern "C" int cn_proposal_density(const cn_density_params* params, const cn_scene* scene, const float* origins,
                                   const float* directions, const float* starts, const float* ends, int64_t num_rays,
                                   int32_t num_samples, float* density, cn_stream_t stream) {
  CN_REQUIRE(params && scene && origins && directions && starts && ends && density, CN_ERR_INVALID,
             "cn_proposal_density: null argument");
  CN_REQUIRE(num_samples > 0, CN_ERR_INVALID, "cn_proposal_density: num_samples must be > 0");
  int rc = cn::validate_grid(params->grid, "proposal grid");
  if (rc) return rc;
  if ((rc = cn::validate_mlp(params->mlp, "proposal mlp", 2 * params->grid.num_levels, 1, 64))) return rc;
  if (num_rays <= 0) return CN_OK;
  cn::DensityDev dp;
  dp.grid = cn::make_grid_dev(params->grid);
  dp.mlp = cn::make_mlp_dev(params->mlp);
  long long nblk = (num_rays * (long long)num_samples + 63) / 64;
  hipLaunchKernelGGL(cn::proposal_density_kernel, dim3(cn::grid_for(nblk, 1, 256 * 16)), dim3(64), 0,
                     cn::as_stream(stream), dp, cn::make_scene_dev(*scene), origins, directions, starts, ends,
                     (long long)num_rays, num_samples, density);
  return cn::check_launch("cn_proposal_density");
}

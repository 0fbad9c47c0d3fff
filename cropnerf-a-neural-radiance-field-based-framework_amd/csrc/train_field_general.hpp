// Shape-generic FruitField backward on the fp32 matrix cores (included by train_field.hip): the training path of the
// field shapes the specialised kernel (train_field_mfma.hpp) is not built for -- fruit_nerf_method_big / _huge
// (geo_feat_dim 30, 3 x 128 semantic layers; fruit_nerf/fruit_nerf_config.py:66-172).
//
// Supported family (every config of the reference): base MLP 2 layers, semantic MLP 2 or 3 layers + Linear(Ht, 1),
// colour MLP 3 layers, all widths <= 128, as long as the per-tile activations fit LDS.  One 512-thread workgroup per CU
// walks 32-sample tiles; activations and deltas live in LDS as [feature][36] with every row count padded to 16 (pad rows
// hold exact zeros); weights are read straight from global memory (L2) as MFMA A operands, with guards instead of
// padding.  Weight and bias gradients are added, tile by tile and without atomics, into a scratch area PRIVATE to the
// workgroup ([workgroups][parameters] floats in the caller's workspace); field_backward_reduce_kernel folds the scratch
// into the gradient tensors afterwards.  Hash-table and appearance-embedding gradients are scatter-adds as in the
// specialised kernel, and so are the optional position / direction gradients for the camera pose refinement.
#pragma once

namespace cn {
namespace gb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TSG = 32, LDG = 36, NTG = 512;
#ifndef CN_GEN_DW_PREFETCH
#define CN_GEN_DW_PREFETCH 4
#endif

struct GenLayer {
  const float* W;
  const float* b;
  int K, N;
  int off_w, off_b;  // offsets (floats) of dW / db inside a workgroup's scratch slice
};

struct GenArgs {
  GenLayer base[2], sem[3], col[3];
  int ns;                    // semantic layers (2 or 3)
  const float* wh;           // semantic head [Ht]
  int off_wh, off_bh;
  int params_per_block;      // floats of one workgroup's scratch slice
  float* scratch;            // [gridDim.x][params_per_block], zeroed by the caller
  // LDS row offsets (rows of LDG floats)
  int r_enc, r_h1, r_g, r_s[3], r_cin, r_c1, r_c2, r_rgb, r_da, r_db, r_dcin, r_dg, r_drgb, r_dsem, rows;
  // field
  const float* table;
  float* g_table;
  const float* emb;
  float* g_emb;
  const float* app_mean;
  GridDev grid;
  int num_levels, geo, app_dim, app_per_camera, sh_unit;
  SceneDev scene;
  const float *origins, *directions, *starts, *ends;
  const int64_t* cam_idx;
  const float *d_density, *d_rgb, *d_sem;
  float *d_pos, *d_dir;  // optional [R*S,3]: gradients w.r.t. the sample position / the ray direction (pose refinement)
  long long R;
  int S;
  CoarseScatter coarse;  // private copies for level 0's gradient (cn_grid.scatter_scratch of the gradient grid)
  CellScatter cells;     // cell-major records of the coarse levels (take precedence for the levels they cover)
  int debug_skip;        // CN_DEBUG_SKIP, timing only: 1 no hash scatter, 4 no weight-gradient products, 64 no forward gathers
};

__device__ __forceinline__ int opaque_i(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// A layer has T = pad16(rows) / 16 row tiles and two 16-sample halves.  With T >= 8 (the 128-wide layers) a wave takes row
// tile `wave` and BOTH halves: its 32 A operands are loaded once and feed two accumulation chains (half the weight loads, half
// the L1 line lookups, one exposed L2 latency per phase instead of two, and two independent matrix chains).  With fewer row
// tiles the 2 T blocks are spread over the waves one (row tile, half) each, as before.
template <bool BOTH>
__device__ __forceinline__ void gen_fwd_blocks(const GenLayer& L, const float* in, float* out, bool relu, int first, int step,
                                               int count, int i, int q, int Kp) {
  for (int blk = first; blk < count; blk += step) {
    const int n0 = (BOTH ? blk : blk >> 1) * 16, s0 = BOTH ? 0 : (blk & 1) * 16;
    f32x4 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * q + r;
      acc0[r] = n < L.N ? L.b[n] : 0.f;
    }
    acc1 = acc0;
    const int nrow = n0 + i;
    // all A operands of the block (one weight row per lane, <= 128 columns) are requested up front: the L2 latency is
    // paid once per 16x16 block instead of once per MFMA
    float areg[32];
    if ((L.K & 3) == 0) {  // 16-byte row loads: a quarter of the load instructions and of the L1's line lookups
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const int k = 16 * kb + 4 * q;
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (nrow < L.N && k < L.K) w = *reinterpret_cast<const f32x4*>(L.W + (size_t)nrow * L.K + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) areg[4 * kb + e] = w[e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const int k = 16 * (j >> 2) + 4 * q + (j & 3);
        areg[j] = (nrow < L.N && k < L.K) ? L.W[(size_t)nrow * L.K + k] : 0.f;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (16 * kb < Kp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* row = in + (16 * kb + 4 * q + e) * LDG + s0 + i;
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[4 * kb + e], row[0], acc0, 0, 0, 0);
          if (BOTH) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[4 * kb + e], row[16], acc1, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v0 = acc0[r], v1 = acc1[r];
      if (relu) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
      }
      out[(n0 + 4 * q + r) * LDG + s0 + i] = v0;
      if (BOTH) out[(n0 + 4 * q + r) * LDG + 16 + i] = v1;
    }
  }
}

// out[n][s] = act(b[n] + sum_k W[n][k] in[k][s]) for all n < pad16(N); rows >= N come out as zeros
__device__ __forceinline__ void gen_fwd(const GenLayer& L, const float* in, float* out, bool relu, int tid) {
  const int lane = opaque_i(tid) & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int Kp = (L.K + 15) & ~15, tiles = ((L.N + 15) & ~15) >> 4;
  if (tiles >= NTG / 64)
    gen_fwd_blocks<true>(L, in, out, relu, wave, NTG / 64, tiles, i, q, Kp);
  else
    gen_fwd_blocks<false>(L, in, out, relu, wave, NTG / 64, 2 * tiles, i, q, Kp);
}

template <bool BOTH>
__device__ __forceinline__ void gen_bwd_blocks(const GenLayer& L, const float* dy, float* dx, const float* mask, int first,
                                               int step, int count, int i, int q, int Np) {
  for (int blk = first; blk < count; blk += step) {
    const int k0 = (BOTH ? blk : blk >> 1) * 16, s0 = BOTH ? 0 : (blk & 1) * 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int kcol = k0 + i;
    float areg[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int n = 16 * (j >> 2) + 4 * q + (j & 3);
      areg[j] = (n < L.N && kcol < L.K) ? L.W[(size_t)n * L.K + kcol] : 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      if (16 * nb < Np) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* row = dy + (16 * nb + 4 * q + e) * LDG + s0 + i;
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[4 * nb + e], row[0], acc0, 0, 0, 0);
          if (BOTH) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[4 * nb + e], row[16], acc1, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + 4 * q + r;
      float v0 = acc0[r], v1 = acc1[r];
      if (mask) {
        v0 = mask[k * LDG + s0 + i] > 0.f ? v0 : 0.f;
        if (BOTH) v1 = mask[k * LDG + 16 + i] > 0.f ? v1 : 0.f;
      }
      dx[k * LDG + s0 + i] = v0;
      if (BOTH) dx[k * LDG + 16 + i] = v1;
    }
  }
}

// dx[k][s] = (sum_n W[n][k] dy[n][s]) * (mask ? mask[k][s] > 0 : 1) for all k < pad16(K)
__device__ __forceinline__ void gen_bwd(const GenLayer& L, const float* dy, float* dx, const float* mask, int tid) {
  const int lane = opaque_i(tid) & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int tiles = ((L.K + 15) & ~15) >> 4, Np = (L.N + 15) & ~15;
  if (tiles >= NTG / 64)
    gen_bwd_blocks<true>(L, dy, dx, mask, wave, NTG / 64, tiles, i, q, Np);
  else
    gen_bwd_blocks<false>(L, dy, dx, mask, wave, NTG / 64, 2 * tiles, i, q, Np);
}

// scratch dW[n][k] += sum_s dy[n][s] x[k][s];  scratch db[n] += sum_s dy[n][s]   (workgroup-private, no atomics)
__device__ __forceinline__ void gen_dw(const GenLayer& L, const float* dy, const float* x, float* scratch, int tid) {
  if (!scratch) return;  // (CN_DEBUG_SKIP bit 4)
  const int lane = opaque_i(tid) & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int Kp = (L.K + 15) & ~15, Np = (L.N + 15) & ~15;
  const int nkt = Kp >> 4, nblocks = (Np >> 4) * nkt;
  // The workgroup's slice of the scratch (206 KB for the big shape, 32 slices to an XCD) does not stay in the L2: a block's
  // read-modify-write waits for the Infinity Cache.  All of a wave's blocks of the layer (<= 8 for widths <= 128) are
  // therefore REQUESTED PF at a time, and added to as their products come out -- PF misses in flight per wave instead of one.
  constexpr int PF = CN_GEN_DW_PREFETCH;
  auto product = [&](int n0, int k0) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sb = 0; sb < TSG; sb += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(dy + (n0 + i) * LDG + sb + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(x + (k0 + i) * LDG + sb + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float av = a[e], bv = b[e];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
      }
    }
    return acc;
  };
#pragma unroll 1
  for (int b0 = wave; b0 < nblocks; b0 += (NTG / 64) * PF) {
    f32x4 pre[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int blk = b0 + (NTG / 64) * j;
      pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (blk < nblocks) {
        const int n0 = (blk / nkt) * 16, k0 = (blk % nkt) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 4 * q + r, k = k0 + i;
          if (n < L.N && k < L.K) pre[j][r] = scratch[L.off_w + n * L.K + k];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int blk = b0 + (NTG / 64) * j;
      if (blk < nblocks) {
        const int n0 = (blk / nkt) * 16, k0 = (blk % nkt) * 16;
        const f32x4 acc = product(n0, k0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 4 * q + r, k = k0 + i;
          if (n < L.N && k < L.K) scratch[L.off_w + n * L.K + k] = pre[j][r] + acc[r];
        }
      }
    }
  }
  for (int n = tid; n < L.N; n += NTG) {
    float sum = 0.f;
    for (int s = 0; s < TSG; ++s) sum += dy[n * LDG + s];
    scratch[L.off_b + n] += sum;
  }
}

__global__ void __launch_bounds__(NTG) field_backward_general_kernel(GenArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x;
  for (int e = tid; e < A.rows * LDG; e += NTG) lds[e] = 0.f;
  float* SCL = lds + A.rows * LDG;  // level records (a per-lane index into the kernarg arrays would go to scratch)
  lds_level_fill(SCL, A.grid, tid);
  __syncthreads();
  float* ENC = lds + A.r_enc * LDG;
  float* H1 = lds + A.r_h1 * LDG;
  float* G = lds + A.r_g * LDG;
  float* CIN = lds + A.r_cin * LDG;
  float* C1 = lds + A.r_c1 * LDG;
  float* C2 = lds + A.r_c2 * LDG;
  float* RGB = lds + A.r_rgb * LDG;
  float* DA = lds + A.r_da * LDG;
  float* DB = lds + A.r_db * LDG;
  float* DCIN = lds + A.r_dcin * LDG;
  float* DG = lds + A.r_dg * LDG;
  float* DRGB = lds + A.r_drgb * LDG;
  float* DSEM = lds + A.r_dsem * LDG;
  float* scratch = (A.debug_skip & 4) ? nullptr : A.scratch + (size_t)blockIdx.x * A.params_per_block;
  const int s = tid & 31, grp = tid >> 5;  // 16 groups of 32 samples
  const int lane = tid & 63;
  const int cin_dim = 16 + A.geo + A.app_dim;
  const long long total = A.R * (long long)A.S;
  const long long ntiles = (total + TSG - 1) / TSG;
  const GenLayer& lastsem = A.sem[A.ns - 1];
  // (one contiguous run of tiles per workgroup: train_field_mfma.hpp on batches sorted by camera and pixel)
  const long long tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
  const long long tile_end = ((long long)blockIdx.x + 1) * tiles_per_wg < ntiles ? ((long long)blockIdx.x + 1) * tiles_per_wg : ntiles;
  for (long long tile = blockIdx.x * tiles_per_wg; tile < tile_end; ++tile) {
    const long long ismp = tile * TSG + s;
    const bool valid = ismp < total;
    const long long ic = valid ? ismp : total - 1;
    const long long r = ic / A.S;
    const float mid = (A.starts[ic] + A.ends[ic]) / 2.f;
    const float dirx = A.directions[3 * r], diry = A.directions[3 * r + 1], dirz = A.directions[3 * r + 2];
    const float wx = A.origins[3 * r] + dirx * mid, wy = A.origins[3 * r + 1] + diry * mid,
                wz = A.origins[3 * r + 2] + dirz * mid;
    float px = wx, py = wy, pz = wz;
    const float self = normalize_position(A.scene, px, py, pz) ? 1.f : 0.f;
    // ---- inputs ---------------------------------------------------------------------------------------------------------
    for (int l = grp; l < A.num_levels; l += 16) {
      const float2 f = (A.debug_skip & 64) ? make_float2(px * 0.01f, py * 0.01f)
                                           : hash_level(A.table, lds_level_rec(SCL, l), A.grid.pos_offset, px, py, pz);
      ENC[(2 * l) * LDG + s] = f.x;
      ENC[(2 * l + 1) * LDG + s] = f.y;
    }
    if (grp == 0) DSEM[s] = valid ? A.d_sem[ic] : 0.f;
    if (grp == 1) {
      float dx = dirx, dy = diry, dz = dirz;
      if (!A.sh_unit) {
        dx = (dx + 1.f) / 2.f;
        dy = (dy + 1.f) / 2.f;
        dz = (dz + 1.f) / 2.f;
      }
      float sh[16];
      sh_deg4(dx, dy, dz, sh);
#pragma unroll
      for (int k = 0; k < 16; ++k) CIN[k * LDG + s] = sh[k];
    }
    {
      const float* a = A.app_per_camera ? A.emb + A.cam_idx[r] * (long long)A.app_dim : A.app_mean;
      for (int k = grp; k < A.app_dim; k += 16) CIN[(16 + A.geo + k) * LDG + s] = a ? a[k] : 0.f;
    }
    __syncthreads();
    // ---- forward ---------------------------------------------------------------------------------------------------------
    gen_fwd(A.base[0], ENC, H1, true, tid);
    __syncthreads();
    gen_fwd(A.base[1], H1, G, false, tid);
    __syncthreads();
    for (int k = grp; k < A.geo; k += 16) CIN[(16 + k) * LDG + s] = G[(1 + k) * LDG + s];
    {
      const float* x = G + LDG;  // geo rows (detached input of the semantic MLP)
      for (int l = 0; l < A.ns; ++l) {
        float* y = lds + A.r_s[l] * LDG;
        gen_fwd(A.sem[l], x, y, l < A.ns - 1, tid);
        __syncthreads();
        x = y;
      }
    }
    gen_fwd(A.col[0], CIN, C1, true, tid);
    __syncthreads();
    gen_fwd(A.col[1], C1, C2, true, tid);
    __syncthreads();
    gen_fwd(A.col[2], C2, RGB, false, tid);
    __syncthreads();
    // ---- semantic branch backward (stops at the detached geo features) ------------------------------------------------------
    {
      const float* sout = lds + A.r_s[A.ns - 1] * LDG;
      const int ht = lastsem.N;
      // head: d_sout[k][s] = wh[k] d_sem[s];  dWh[k] += sum_s d_sem[s] sout[k][s];  dbh += sum_s d_sem[s]
      for (int k = grp; k < ((ht + 15) & ~15); k += 16) DA[k * LDG + s] = k < ht ? A.wh[k] * DSEM[s] : 0.f;
      for (int k = tid; k < ht && scratch; k += NTG) {
        float sum = 0.f;
        for (int j = 0; j < TSG; ++j) sum += DSEM[j] * sout[k * LDG + j];
        scratch[A.off_wh + k] += sum;
      }
      if (tid == 0 && scratch) {
        float sum = 0.f;
        for (int j = 0; j < TSG; ++j) sum += DSEM[j];
        scratch[A.off_bh] += sum;
      }
      __syncthreads();
      float* dcur = DA;
      float* dnext = DB;
      for (int l = A.ns - 1; l >= 0; --l) {
        const float* xin = l == 0 ? G + LDG : lds + A.r_s[l - 1] * LDG;
        gen_dw(A.sem[l], dcur, xin, scratch, tid);
        if (l > 0) gen_bwd(A.sem[l], dcur, dnext, xin, tid);  // gate: the input is a post-ReLU activation
        __syncthreads();
        float* t = dcur;
        dcur = dnext;
        dnext = t;
      }
    }
    // ---- colour branch backward -------------------------------------------------------------------------------------------
    if (tid < 96) {  // d_rgb_pre = d_rgb * rgb * (1 - rgb), rows 0..2 (rows 3..15 of DRGB stay zero)
      const int row = tid >> 5;
      const float sg = 1.f / (1.f + expf(-RGB[row * LDG + s]));
      DRGB[row * LDG + s] = valid ? A.d_rgb[3 * ic + row] * sg * (1.f - sg) : 0.f;
    }
    __syncthreads();
    gen_dw(A.col[2], DRGB, C2, scratch, tid);
    gen_bwd(A.col[2], DRGB, DA, C2, tid);
    __syncthreads();
    gen_dw(A.col[1], DA, C1, scratch, tid);
    gen_bwd(A.col[1], DA, DB, C1, tid);
    __syncthreads();
    gen_dw(A.col[0], DB, CIN, scratch, tid);
    gen_bwd(A.col[0], DB, DCIN, nullptr, tid);
    __syncthreads();
    // ---- d(base output): row 0 = density logit through trunc_exp and the selector, rows 1..geo from the colour input ----
    for (int row = grp; row < 32; row += 16) {
      float v = 0.f;
      if (row == 0) {
        const float dd = valid ? A.d_density[ic] : 0.f;
        v = dd * self * expf(fminf(fmaxf(G[s], -15.f), 15.f));
      } else if (row <= A.geo) {
        v = DCIN[(15 + row) * LDG + s];
      }
      DG[row * LDG + s] = v;
    }
    if (A.d_dir && grp == 2 && valid) {
      float gsh[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) gsh[k] = DCIN[k * LDG + s];
      float dx = dirx, dy = diry, dz = dirz;
      const float chain = A.sh_unit ? 1.f : 0.5f;
      if (!A.sh_unit) {
        dx = (dx + 1.f) / 2.f;
        dy = (dy + 1.f) / 2.f;
        dz = (dz + 1.f) / 2.f;
      }
      float gx, gy, gz;
      sh_deg4_backward(dx, dy, dz, gsh, gx, gy, gz);
      A.d_dir[3 * ismp] = gx * chain;
      A.d_dir[3 * ismp + 1] = gy * chain;
      A.d_dir[3 * ismp + 2] = gz * chain;
    }
    if (A.app_per_camera) {
      // a tile's samples belong to one ray or to a few consecutive ones (one camera row each): sum the runs of equal ray
      // inside every 16-lane row first; the last lane of a run adds the sums (per-sample atomics would be up to 32-way
      // conflicting requests on the same addresses from every tile of every workgroup)
      for (int k0 = 0; k0 < A.app_dim; k0 += 32) {  // uniform trip count: every lane runs the DPP ops
        const int ka = k0 + grp, kb = k0 + 16 + grp;
        float ga = (valid && ka < A.app_dim) ? DCIN[(16 + A.geo + ka) * LDG + s] : 0.f;
        float gb = (valid && kb < A.app_dim) ? DCIN[(16 + A.geo + kb) * LDG + s] : 0.f;
        const bool last = row_run_reduce(valid ? (unsigned)r : 0xffffffffu, ga, gb, tid & 15);
        if (last && valid) {
          float* ge = A.g_emb + A.cam_idx[r] * (long long)A.app_dim;
          if (ka < A.app_dim && ga != 0.f) cn_atomic_add(ge + ka, ga);
          if (kb < A.app_dim && gb != 0.f) cn_atomic_add(ge + kb, gb);
        }
      }
    }
    __syncthreads();
    gen_dw(A.base[1], DG, H1, scratch, tid);
    gen_bwd(A.base[1], DG, DA, H1, tid);
    __syncthreads();
    gen_dw(A.base[0], DA, ENC, scratch, tid);
    gen_bwd(A.base[0], DA, DB, nullptr, tid);
    __syncthreads();
    // ---- hash-table gradient (16 consecutive lanes = 16 consecutive samples of one level: run-length pre-reduction) ----
    float gpx = 0.f, gpy = 0.f, gpz = 0.f;
    for (int l0 = 0; l0 < A.num_levels; l0 += 16) {
      const int l = l0 + grp;
      const bool on = l < A.num_levels;
      const int lc = on ? l : 0;
      const bool sc_on = on && valid && !(A.debug_skip & 1);
      const float g0 = sc_on ? DB[(2 * lc) * LDG + s] : 0.f, g1 = sc_on ? DB[(2 * lc + 1) * LDG + s] : 0.f;
      // (a wave holds two levels, 32 lanes each; the branch splits it along whole 16-lane rows)
      if (on && lc < A.cells.num_levels) {
        // cell-major level: C1 / C2 / CIN / DCIN are dead by now and host the 16-lane hand-over of two waves each (the host
        // enables the path only when they are large enough)
        const int wv = tid >> 6;
        float* tb = (wv < 2 ? C1 : wv < 4 ? C2 : wv < 6 ? CIN : DCIN) + (wv & 1) * (64 * 17);
        const unsigned nl = cell_n_of(A.cells, lc);
        float* rec = A.cells.base + cell_offset_of(A.cells, lc) +
                     (size_t)(blockIdx.x % cell_copies_of(A.cells, lc)) * ((size_t)nl * nl * nl * 16);
        if (A.d_pos)
          hash_level_backward_cells<true>(rec, nl, tb, A.g_table, A.table, lds_level_rec(SCL, lc), A.grid.pos_offset, px, py,
                                          pz, g0, g1, lane, gpx, gpy, gpz);
        else
          hash_level_backward_cells<false>(rec, nl, tb, A.g_table, A.table, lds_level_rec(SCL, lc), A.grid.pos_offset, px, py,
                                           pz, g0, g1, lane, gpx, gpy, gpz);
      } else if (on && lc == 0 && A.coarse.base) {
        float* mine = A.coarse.base + (size_t)(blockIdx.x % A.coarse.copies) * (2u * A.coarse.n1 * A.coarse.n1 * A.coarse.n1);
        if (A.d_pos)
          hash_level_backward_private<true>(mine, A.coarse.n1, A.g_table, A.table, lds_level_rec(SCL, 0), A.grid.pos_offset,
                                            px, py, pz, g0, g1, lane, gpx, gpy, gpz);
        else
          hash_level_backward_private<false>(mine, A.coarse.n1, A.g_table, A.table, lds_level_rec(SCL, 0), A.grid.pos_offset,
                                             px, py, pz, g0, g1, lane, gpx, gpy, gpz);
      } else if (A.d_pos)
        hash_level_backward<true>(A.g_table, A.table, lds_level_rec(SCL, lc), A.grid.pos_offset, px, py, pz, g0, g1,
                                  lane, gpx, gpy, gpz);
      else
        hash_level_backward<false>(A.g_table, A.table, lds_level_rec(SCL, lc), A.grid.pos_offset, px, py, pz, g0,
                                   g1, lane, gpx, gpy, gpz);
    }
    if (A.d_pos) {  // per-group partials -> LDS (DA is dead by now) -> one thread per sample sums the 16 groups
      __syncthreads();
      DA[(3 * grp + 0) * LDG + s] = gpx;
      DA[(3 * grp + 1) * LDG + s] = gpy;
      DA[(3 * grp + 2) * LDG + s] = gpz;
      __syncthreads();
      if (grp == 0 && valid) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
          gx += DA[(3 * l + 0) * LDG + s];
          gy += DA[(3 * l + 1) * LDG + s];
          gz += DA[(3 * l + 2) * LDG + s];
        }
        normalize_position_backward(A.scene, wx, wy, wz, self, gx, gy, gz);
        A.d_pos[3 * ismp] = gx;
        A.d_pos[3 * ismp + 1] = gy;
        A.d_pos[3 * ismp + 2] = gz;
      }
    }
    __syncthreads();
  }
}

// grads[e] += sum over workgroups of scratch[b][e] for one parameter tensor at scratch offset `off`
__global__ void __launch_bounds__(256)
field_backward_reduce_kernel(const float* __restrict__ scratch, int nblocks, int params_per_block, int off, int n,
                             float* __restrict__ grad) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    float sum = 0.f;
    for (int b = 0; b < nblocks; ++b) sum += scratch[(size_t)b * params_per_block + off + e];
    grad[e] += sum;
  }
}

// The same for EVERY parameter tensor in one launch (18 launches of the kernel above were 1.5 ms of a _big iteration: a thread
// walked all the workgroups' slices by itself).  A workgroup takes 32 consecutive scratch elements; its 8 groups of 32 threads
// take every 8th slice each (128-byte rows, coalesced) and the partial sums meet in LDS in a fixed order.
struct ReduceTargets {
  float* g[24];
  int off[24], n[24];
  int count;
};
__global__ void __launch_bounds__(256)
field_backward_reduce_all_kernel(const float* __restrict__ scratch, int nblocks, int params_per_block, ReduceTargets T) {
  __shared__ float part[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float sum = 0.f;
  if (e < params_per_block)
    for (int b = g; b < nblocks; b += 8) sum += scratch[(size_t)b * params_per_block + e];
  part[g][threadIdx.x & 31] = sum;
  __syncthreads();
  if (g == 0 && e < params_per_block) {
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += part[k][threadIdx.x];
    for (int t = 0; t < T.count; ++t)
      if (e >= T.off[t] && e < T.off[t] + T.n[t]) T.g[t][e - T.off[t]] += total;
  }
}

}  // namespace gb
}  // namespace cn

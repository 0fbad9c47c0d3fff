// Training, ray side: losses + backward of the volume renderer and of the interlevel (proposal) loss.
//   cn_train_render_backward  : FruitModel.get_loss_dict rgb + semantics terms (fruit_nerf/fruit_nerf.py:601-608) on top of
//                               get_weights + RGBRenderer("last_sample", training) + SemanticRenderer (:556-591), forward
//                               and backward in one wave-per-ray kernel.
//   cn_interlevel_backward    : interlevel_loss term (:609-612; nerfstudio losses.interlevel_loss / lossfun_outer / outer)
//                               of one proposal level: loss and d(loss)/d(proposal density).
//   cn_adam_step              : torch.optim.Adam update (fruit_nerf/fruit_nerf_config.py:45-60), elementwise.
//   cn_radam_step             : torch.optim.RAdam update (the _big / _huge methods, fruit_nerf_config.py:101-167).
// Gradients w.r.t. per-sample field outputs leave as [R,S] arrays; train_field.hip turns them into parameter gradients.
#include "composite_dev.hpp"
#include "cn_det.hpp"

namespace cn {

constexpr int TRAIN_MAX_S = 512;

// reverse-order helpers: suffix sums are prefix sums of the lane-reversed chunk
__device__ __forceinline__ float wave_total(float v) { return wave_read(wave_inclusive_scan(v), 63); }

// one wave per ray.  LDS per wave: w[S] | P[S] (exclusive prefix of delta*sigma)
__global__ void __launch_bounds__(256)
train_render_backward_kernel(const float* __restrict__ starts, const float* __restrict__ ends,
                             const float* __restrict__ density, const float* __restrict__ rgb,
                             const float* __restrict__ sem, const float* __restrict__ image,
                             const float* __restrict__ mask, long long R, int S, float sem_weight,
                             float* __restrict__ out_rgb, float* __restrict__ out_sem, float* __restrict__ out_acc,
                             float* __restrict__ out_w, float* __restrict__ d_density, float* __restrict__ d_rgb,
                             float* __restrict__ d_sem, float* __restrict__ loss_sums,
                             const float* __restrict__ spacing_bins) {
  extern __shared__ __align__(16) float lds[];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  float* wbuf = lds + wave * 2 * S;
  float* pbuf = wbuf + S;
  const long long waves = (long long)gridDim.x * 4;
  const float inv_3r = 1.f / (3.f * (float)R), inv_r = 1.f / (float)R;
  float mse_sum = 0.f, bce_sum = 0.f;  // this wave's rays (lane 0): one atomic per wave at the end, not one per ray on one address
  float dist_sum = 0.f;
  for (long long r = blockIdx.x * 4LL + wave; r < R; r += waves) {
    const long long base = r * (long long)S;
    // ---- forward: weights, rgb / accumulation / semantics ------------------------------------------------------
    float carry = 0.f, ar = 0.f, ag = 0.f, ab = 0.f, aw = 0.f, as = 0.f;
    // nerfstudio's distortion_loss of this ray (the "distortion" metric, fruit_nerf.py:643) rides along when the caller hands
    // over the spacing bins: sum_ij w_i w_j |u_i - u_j| = 2 sum_i w_i (u_i W_<i - (w u)_<i) for ascending mid-points u -- two
    // more prefix sums of the scan that is here anyway, O(S) instead of the S^2 products of cn_distortion_metric; u is taken
    // relative to the ray's first bin edge, so the difference of the two prefix terms does not cancel at the far end of [0, 1]
    float dW = 0.f, dWU = 0.f, dterm = 0.f;
    const float u_ref = spacing_bins ? spacing_bins[r * (S + 1)] : 0.f;
    for (int c0 = 0; c0 < S; c0 += 64) {
      const int i = c0 + lane;
      const bool valid = i < S;
      const int ic = valid ? i : S - 1;
      float dd = valid ? (ends[base + ic] - starts[base + ic]) * density[base + ic] : 0.f;
      float incl = wave_inclusive_scan(dd);
      float P = carry + (incl - dd);
      float w = valid ? nan_to_num((1.f - expf(-dd)) * expf(-P)) : 0.f;
      carry += wave_read(incl, 63);
      if (spacing_bins) {  // (wave-uniform)
        const float b0 = spacing_bins[r * (S + 1) + ic], b1 = spacing_bins[r * (S + 1) + ic + 1];
        const float u = (b0 + b1) / 2.f - u_ref, wu = w * u;
        const float sw = wave_inclusive_scan(w), swu = wave_inclusive_scan(wu);
        dterm += 2.f * w * (u * (dW + sw - w) - (dWU + swu - wu)) + w * w * (b1 - b0) / 3.f;
        dW += wave_read(sw, 63);
        dWU += wave_read(swu, 63);
      }
      if (valid) {
        wbuf[i] = w;
        pbuf[i] = P;
        if (out_w) out_w[base + i] = w;
      }
      ar += w * rgb[3 * (base + ic) + 0];
      ag += w * rgb[3 * (base + ic) + 1];
      ab += w * rgb[3 * (base + ic) + 2];
      aw += w;
      as += w * sem[base + ic];
    }
    const float acc = wave_sum(aw);
    const float lr_ = rgb[3 * (base + S - 1) + 0], lg_ = rgb[3 * (base + S - 1) + 1], lb_ = rgb[3 * (base + S - 1) + 2];
    const float cr = wave_sum(ar) + lr_ * (1.f - acc);
    const float cg = wave_sum(ag) + lg_ * (1.f - acc);
    const float cb = wave_sum(ab) + lb_ * (1.f - acc);
    const float so = wave_sum(as);
    if (spacing_bins) {
      const float t = wave_sum(dterm);
      if (lane == 0) dist_sum += t;
    }
    // ---- losses and their derivatives w.r.t. the rendered values ----------------------------------------------------
    const float e0 = cr - image[3 * r + 0], e1 = cg - image[3 * r + 1], e2 = cb - image[3 * r + 2];
    const float y = mask[r];
    const float G0 = 2.f * e0 * inv_3r, G1 = 2.f * e1 * inv_3r, G2 = 2.f * e2 * inv_3r;  // d MSE / d rgb
    const float gs = sem_weight * (sigmoidf(so) - y) * inv_r;                           // d BCE / d semantics
    if (lane == 0) {
      if (out_rgb) {
        out_rgb[3 * r + 0] = cr;
        out_rgb[3 * r + 1] = cg;
        out_rgb[3 * r + 2] = cb;
      }
      if (out_sem) out_sem[r] = so;
      if (out_acc) out_acc[r] = acc;
      mse_sum += e0 * e0 + e1 * e1 + e2 * e2;
      bce_sum += fmaxf(so, 0.f) - so * y + log1pf(expf(-fabsf(so)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- backward, chunks in reverse order: d dd_j = gw_j (T_j - w_j) - sum_{i>j} gw_i w_i -----------------------
    float suffix = 0.f;  // sum over later chunks of gw_i * w_i
    for (int c0 = ((S - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
      const int i = c0 + (63 - lane);  // lanes walk the chunk backwards, so a prefix scan is a suffix sum
      const bool valid = i < S;
      const int ic = valid ? i : S - 1;
      const float w = valid ? wbuf[ic] : 0.f;
      const float c_r = rgb[3 * (base + ic) + 0], c_g = rgb[3 * (base + ic) + 1], c_b = rgb[3 * (base + ic) + 2];
      const float gw = G0 * (c_r - lr_) + G1 * (c_g - lg_) + G2 * (c_b - lb_);
      const float term = valid ? gw * w : 0.f;
      const float incl = wave_inclusive_scan(term);
      const float later = suffix + (incl - term);  // strictly later samples
      suffix += wave_read(incl, 63);
      if (valid) {
        const float T = expf(-pbuf[ic]);
        const float delta = ends[base + ic] - starts[base + ic];
        d_density[base + ic] = delta * (gw * (T - w) - later);
        const float k = (ic == S - 1) ? w + (1.f - acc) : w;
        d_rgb[3 * (base + ic) + 0] = G0 * k;
        d_rgb[3 * (base + ic) + 1] = G1 * k;
        d_rgb[3 * (base + ic) + 2] = G2 * k;
        d_sem[base + ic] = gs * w;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // one atomic pair per workgroup (every ray's own atomic on these two addresses is served one after the other: at
  // 4096 rays that alone was a third of this kernel)
  __shared__ float red[4][3];
  if (lane == 0) {
    red[wave][0] = mse_sum;
    red[wave][1] = bce_sum;
    red[wave][2] = dist_sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = red[0][0] + red[1][0] + red[2][0] + red[3][0], b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    if (a != 0.f || b != 0.f) {
      cn_atomic_add(loss_sums + 0, a);
      cn_atomic_add(loss_sums + 1, b);
    }
    const float c = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    if (spacing_bins && c != 0.f) cn_atomic_add(loss_sums + 4, c);
  }
}

// first index in [0,n) with a[idx] > v   (torch.searchsorted side="right")
__device__ __forceinline__ int upper_bound(const float* a, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// one wave per ray.  LDS per wave: cp[Sp+1] | wp[Sp] | P[Sp] | cy[Sp+1] | diff[Sp+1]
__device__ __forceinline__ void
interlevel_backward_rays(float* lds, const float* __restrict__ c_bins, const float* __restrict__ w_final,
                         const float* __restrict__ cp_bins, const float* __restrict__ starts_p,
                         const float* __restrict__ ends_p, const float* __restrict__ density_p, long long R, int Sf,
                         int Sp, float mult, float* __restrict__ d_density_p, float* __restrict__ loss_sum) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int stride = 5 * Sp + 3;
  float* cp = lds + wave * stride;
  float* wp = cp + Sp + 1;
  float* Pp = wp + Sp;
  float* cy = Pp + Sp;
  float* diff = cy + Sp + 1;
  const long long waves = (long long)gridDim.x * 4;
  const float eps = 1.0e-7f;
  float loss_acc = 0.f;  // this wave's rays: one atomic per wave
  for (long long r = blockIdx.x * 4LL + wave; r < R; r += waves) {
    const long long bp = r * (long long)Sp, bf = r * (long long)Sf;
    for (int e = lane; e <= Sp; e += 64) {
      cp[e] = cp_bins[r * (Sp + 1) + e];
      diff[e] = 0.f;
    }
    // proposal weights, their exclusive dd prefix and their cumulative sum cy[m] = sum_{k<m} wp_k
    float carry = 0.f, cw = 0.f;
    for (int c0 = 0; c0 < Sp; c0 += 64) {
      const int i = c0 + lane;
      const bool valid = i < Sp;
      const int ic = valid ? i : Sp - 1;
      float dd = valid ? (ends_p[bp + ic] - starts_p[bp + ic]) * density_p[bp + ic] : 0.f;
      float incl = wave_inclusive_scan(dd);
      float P = carry + (incl - dd);
      float w = valid ? nan_to_num((1.f - expf(-dd)) * expf(-P)) : 0.f;
      carry += wave_read(incl, 63);
      float wi = wave_inclusive_scan(w);
      if (valid) {
        wp[i] = w;
        Pp[i] = P;
        cy[i + 1] = cw + wi;
      }
      cw += wave_read(wi, 63);
    }
    if (lane == 0) cy[0] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // loss terms of the final samples; d loss / d wp as range adds on a difference array
    float lsum = 0.f;
    for (int i = lane; i < Sf; i += 64) {
      const float t0s = c_bins[r * (Sf + 1) + i], t0e = c_bins[r * (Sf + 1) + i + 1];
      const float w = w_final[bf + i];
      int lo = upper_bound(cp, Sp, t0s) - 1;       // t1_starts = cp[0..Sp)
      int hi = upper_bound(cp + 1, Sp, t0e);       // t1_ends   = cp[1..Sp]
      lo = min(max(lo, 0), Sp - 1);
      hi = min(max(hi, 0), Sp - 1);
      const float outer = cy[hi + 1] - cy[lo];
      const float dv = w - outer;
      if (dv > 0.f) {
        lsum += dv * dv / (w + eps);
        const float g = -2.f * dv / (w + eps) * mult;  // d/d outer, scaled by loss_mult / (R*Sf)
        atomicAdd(diff + lo, g);
        atomicAdd(diff + hi + 1, -g);
      }
    }
    loss_acc += wave_sum(lsum);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // gw[m] = prefix(diff)[m]  (in place, chunked scan)
    float run = 0.f;
    for (int c0 = 0; c0 < Sp; c0 += 64) {
      const int i = c0 + lane;
      float v = i < Sp ? diff[i] : 0.f;
      float incl = wave_inclusive_scan(v);
      if (i < Sp) diff[i] = run + incl;
      run += wave_read(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // weights backward of the proposal level
    float suffix = 0.f;
    for (int c0 = ((Sp - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
      const int i = c0 + (63 - lane);
      const bool valid = i < Sp;
      const int ic = valid ? i : Sp - 1;
      const float w = valid ? wp[ic] : 0.f;
      const float gw = valid ? diff[ic] : 0.f;
      const float term = gw * w;
      const float incl = wave_inclusive_scan(term);
      const float later = suffix + (incl - term);
      suffix += wave_read(incl, 63);
      if (valid) {
        const float T = expf(-Pp[ic]);
        d_density_p[bp + ic] = (ends_p[bp + ic] - starts_p[bp + ic]) * (gw * (T - w) - later);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __shared__ float red[4];  // one atomic per workgroup
  if (lane == 0) red[wave] = loss_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = red[0] + red[1] + red[2] + red[3];
    if (a != 0.f) cn_atomic_add(loss_sum, a);
  }
}
__global__ void __launch_bounds__(256)
interlevel_backward_kernel(const float* __restrict__ c_bins, const float* __restrict__ w_final,
                           const float* __restrict__ cp_bins, const float* __restrict__ starts_p,
                           const float* __restrict__ ends_p, const float* __restrict__ density_p, long long R, int Sf,
                           int Sp, float mult, float* __restrict__ d_density_p, float* __restrict__ loss_sum) {
  extern __shared__ __align__(16) float lds[];
  interlevel_backward_rays(lds, c_bins, w_final, cp_bins, starts_p, ends_p, density_p, R, Sf, Sp, mult, d_density_p, loss_sum);
}
// every proposal level of an iteration in ONE launch (blockIdx.y = level): the levels only share read-only inputs, and at the
// reference's 4 096-ray batches each of them is a launch-latency-sized kernel
constexpr int INTERLEVEL_MAX_LEVELS = 4;
struct InterlevelLevels {
  const float *cp_bins[INTERLEVEL_MAX_LEVELS], *starts[INTERLEVEL_MAX_LEVELS], *ends[INTERLEVEL_MAX_LEVELS],
      *density[INTERLEVEL_MAX_LEVELS];
  float* d_density[INTERLEVEL_MAX_LEVELS];
  int Sp[INTERLEVEL_MAX_LEVELS];
};
__global__ void __launch_bounds__(256)
interlevel_backward_levels_kernel(const float* __restrict__ c_bins, const float* __restrict__ w_final, InterlevelLevels L,
                                  long long R, int Sf, float mult, float* __restrict__ loss_sum) {
  extern __shared__ __align__(16) float lds[];
  const float *cp = L.cp_bins[0], *st = L.starts[0], *en = L.ends[0], *de = L.density[0];
  float* dd = L.d_density[0];
  int Sp = L.Sp[0];
#pragma unroll
  for (int k = 1; k < INTERLEVEL_MAX_LEVELS; ++k) {  // (block-uniform selects: no run-time index into the kernel arguments)
    const bool m = (int)blockIdx.y == k;
    cp = m ? L.cp_bins[k] : cp;
    st = m ? L.starts[k] : st;
    en = m ? L.ends[k] : en;
    de = m ? L.density[k] : de;
    dd = m ? L.d_density[k] : dd;
    Sp = m ? L.Sp[k] : Sp;
  }
  interlevel_backward_rays(lds, c_bins, w_final, cp, st, en, de, R, Sf, Sp, mult, dd, loss_sum);
}

// nerfstudio distortion_loss on the final level (a metric in the reference, fruit_nerf.py:643):
// mean over rays of  sum_ij w_i w_j |u_i - u_j| + 1/3 sum_i w_i^2 (c_{i+1} - c_i),  u = bin mid-points in the spacing domain.
// One wave per ray; LDS per wave: w[S] | u[S].
__global__ void __launch_bounds__(256)
distortion_kernel(const float* __restrict__ bins, const float* __restrict__ weights, long long R, int S,
                  float* __restrict__ sum_out) {
  extern __shared__ __align__(16) float lds[];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  float* w = lds + wave * 2 * S;
  float* u = w + S;
  const long long waves = (long long)gridDim.x * 4;
  float acc_total = 0.f;  // this wave's rays: one atomic per wave
  for (long long r = blockIdx.x * 4LL + wave; r < R; r += waves) {
    float intra = 0.f;
    for (int i = lane; i < S; i += 64) {
      const float c0 = bins[r * (S + 1) + i], c1 = bins[r * (S + 1) + i + 1];
      const float wi = weights[r * (long long)S + i];
      w[i] = wi;
      u[i] = (c0 + c1) / 2.f;
      intra += wi * wi * (c1 - c0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float inter = 0.f;
    for (int i = lane; i < S; i += 64) {
      float s = 0.f;
      const float ui = u[i];
      for (int j = 0; j < S; ++j) s += w[j] * fabsf(ui - u[j]);
      inter += w[i] * s;
    }
    acc_total += wave_sum(inter + intra / 3.f);
    __builtin_amdgcn_wave_barrier();
  }
  __shared__ float red[4];  // one atomic per workgroup
  if (lane == 0) red[wave] = acc_total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = red[0] + red[1] + red[2] + red[3];
    if (a != 0.f) cn_atomic_add(sum_out, a);
  }
}

// Camera pose refinement, backward.  Sample positions are o + d * mid with constant mid (the sampler's bins are
// detached), so per ray  dL/do = sum_s dL/dp_s  and  dL/dd = sum_s mid_s dL/dp_s (+ the SH-input gradient of the colour
// branch).  One wave per ray; the sums are ACCUMULATED into d_origins / d_directions [R,3].
__global__ void __launch_bounds__(256)
ray_backward_kernel(const float* __restrict__ d_pos, const float* __restrict__ d_dir_samples,
                    const float* __restrict__ starts, const float* __restrict__ ends, long long R, int S,
                    float* __restrict__ d_origins, float* __restrict__ d_directions) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const long long waves = (long long)gridDim.x * 4;
  for (long long r = blockIdx.x * 4LL + wave; r < R; r += waves) {
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    for (int i = lane; i < S; i += 64) {
      const long long k = r * (long long)S + i;
      const float mid = (starts[k] + ends[k]) / 2.f;
      const float gx = d_pos[3 * k], gy = d_pos[3 * k + 1], gz = d_pos[3 * k + 2];
      ox += gx;
      oy += gy;
      oz += gz;
      dx = fmaf(mid, gx, dx);
      dy = fmaf(mid, gy, dy);
      dz = fmaf(mid, gz, dz);
      if (d_dir_samples) {
        dx += d_dir_samples[3 * k];
        dy += d_dir_samples[3 * k + 1];
        dz += d_dir_samples[3 * k + 2];
      }
    }
    ox = wave_sum(ox); oy = wave_sum(oy); oz = wave_sum(oz);
    dx = wave_sum(dx); dy = wave_sum(dy); dz = wave_sum(dz);
    if (lane == 0) {
      d_origins[3 * r] += ox;
      d_origins[3 * r + 1] += oy;
      d_origins[3 * r + 2] += oz;
      d_directions[3 * r] += dx;
      d_directions[3 * r + 1] += dy;
      d_directions[3 * r + 2] += dz;
    }
  }
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// Backward of exp_map_SO3xR3 applied to a ray (o' = o + t, d' = R(w) d with R = I + f1 K + f2 K^2, K = skew(w),
// theta = sqrt(max(|w|^2, 1e-4)), f1 = sin(theta)/theta, f2 = (1 - cos(theta))/theta^2):
//   dL/dt = dL/do'
//   dL/dw = f1 (d x g) + f2 ((w x d) x g + d x (g x w)) + [|w|^2 >= 1e-4] (w / theta) (f1' g.(w x d) + f2' g.(w x (w x d)))
// with g = dL/dd'.  One thread per ray, atomics into grad_pose[camera].
// LDS = true: the workgroup sums its rays' contributions per camera in LDS ([C][6] floats, ds_add_f32) and adds every
// non-zero entry to grad_pose once -- the rays' 6 R atomics all land on the C x 24 bytes of grad_pose (38 lines for 100
// cameras: requests to one line are served one after the other, 0.19 ms at 65 536 rays).
template <bool LDS>
__global__ void __launch_bounds__(256)
pose_backward_kernel(const float* __restrict__ adj, const int64_t* __restrict__ cam, const float* __restrict__ dirs_raw,
                     const float* __restrict__ d_origins, const float* __restrict__ d_directions, long long R, int C,
                     float* __restrict__ grad_pose) {
  extern __shared__ __align__(16) float pose_acc[];
  if (LDS) {
    for (int e = threadIdx.x; e < 6 * C; e += blockDim.x) pose_acc[e] = 0.f;
    __syncthreads();
  }
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < R; r += (long long)gridDim.x * blockDim.x) {
    const long long c = cam[r];
    const float w[3] = {adj[6 * c + 3], adj[6 * c + 4], adj[6 * c + 5]};
    const float d[3] = {dirs_raw[3 * r], dirs_raw[3 * r + 1], dirs_raw[3 * r + 2]};
    const float g[3] = {d_directions[3 * r], d_directions[3 * r + 1], d_directions[3 * r + 2]};
    const float nrm = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const float th = sqrtf(fmaxf(nrm, 1e-4f));
    const float inv = 1.f / th;
    const float sn = sinf(th), cs = cosf(th);
    const float f1 = inv * sn, f2 = inv * inv * (1.f - cs);
    float dxg[3], wxd[3], t1[3], gxw[3], t2[3], wwd[3];
    cross3(d, g, dxg);
    cross3(w, d, wxd);
    cross3(wxd, g, t1);
    cross3(g, w, gxw);
    cross3(d, gxw, t2);
    cross3(w, wxd, wwd);
    float radial = 0.f;
    if (nrm >= 1e-4f) {
      const float df1 = (th * cs - sn) * inv * inv;
      const float df2 = (th * sn - 2.f * (1.f - cs)) * inv * inv * inv;
      const float a = g[0] * wxd[0] + g[1] * wxd[1] + g[2] * wxd[2];
      const float b = g[0] * wwd[0] + g[1] * wwd[1] + g[2] * wwd[2];
      radial = (df1 * a + df2 * b) * inv;
    }
    float* dst = LDS ? pose_acc + 6 * c : grad_pose + 6 * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float go = d_origins[3 * r + k], gr = f1 * dxg[k] + f2 * (t1[k] + t2[k]) + radial * w[k];
      if (LDS) {
        atomicAdd(dst + k, go);
        atomicAdd(dst + 3 + k, gr);
      } else {
        cn_atomic_add(dst + k, go);
        cn_atomic_add(dst + 3 + k, gr);
      }
    }
  }
  if (LDS) {
    __syncthreads();
    for (int e = threadIdx.x; e < 6 * C; e += blockDim.x) {
      const float v = pose_acc[e];
      if (v != 0.f) cn_atomic_add(grad_pose + e, v);
    }
  }
}

// camera_opt_regularizer (nerfstudio CameraOptimizer.get_loss_dict): mean_c |t_c| * trans + mean_c |w_c| * rot.
// Adds the loss to *loss_out and its gradient to grad_pose (subgradient 0 at a zero vector, like torch's norm).
__global__ void __launch_bounds__(256)
pose_regularizer_kernel(const float* __restrict__ adj, int C, float trans, float rot, float* __restrict__ grad_pose,
                        float* __restrict__ loss_out) {
  float local = 0.f;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* v = adj + 6 * c + 3 * h;
      const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      const float pen = (h ? rot : trans) / (float)C;
      local += n * pen;
      if (grad_pose && n > 0.f) {
        for (int k = 0; k < 3; ++k) grad_pose[6 * c + 3 * h + k] += pen * v[k] / n;
      }
    }
  }
  local = wave_sum(local);
  if (lane_id() == 0 && local != 0.f) cn_atomic_add(loss_out, local);
}

// get_loss_dict (fruit_nerf.py:601-615) and the scalar part of get_metrics_dict (:639-645) from the kernels' loss sums, in ONE
// launch (the host composed them from ~12 one-element ATen kernels per iteration): out[0] rgb_loss = sum / (3 R); [1]
// semantics_loss = weight * sum / R; [2] interlevel_loss = mult * sum / (R S); [3] camera_opt_regularizer (already a mean);
// [4] psnr = -10 log10(rgb_loss); [5] / [6] the Frobenius norms of the translation / rotation halves of pose_adjustment
// (CameraOptimizer.get_metrics_dict); [7] the distortion metric = sums[4] / R when the caller accumulated cn_distortion_metric's
// sum there (five sums), else 0.
__global__ void __launch_bounds__(64)
train_epilogue_kernel(const float* __restrict__ sums, double inv_3r, double sem_over_r, double inter_over_rs, double inv_r,
                      int with_distortion, const float* __restrict__ pose, int C, float* __restrict__ out) {
  float t2 = 0.f, r2 = 0.f;
  if (pose)
    for (int c = threadIdx.x; c < C; c += 64) {
      const float* v = pose + 6 * c;
      t2 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      r2 += v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
    }
  t2 = wave_sum(t2);
  r2 = wave_sum(r2);
  if (threadIdx.x == 0) {
    const float rgb = (float)((double)sums[0] * inv_3r);
    out[0] = rgb;
    out[1] = (float)((double)sums[1] * sem_over_r);
    out[2] = (float)((double)sums[2] * inter_over_rs);
    out[3] = sums[3];
    out[4] = -10.f * log10f(rgb);
    out[5] = sqrtf(t2);
    out[6] = sqrtf(r2);
    out[7] = with_distortion ? (float)((double)sums[4] * inv_r) : 0.f;
  }
}

// One Adam update, every rounding spelled out: the three kernels below (scalars as arguments, scalars in device memory, several
// groups per launch) must produce the same bits, and the compiler's choice of which product to contract into an FMA differs
// with where the scalars live.  (1 - beta arrives rounded from double on the host, like torch's.)
__device__ __forceinline__ void adam_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, long long i,
                                            float gi, float step_size, float beta1, float beta2, float omb1, float omb2,
                                            float inv_sqrt_bc2, float eps) {
  const float mi = __fmaf_rn(beta1, m[i], __fmul_rn(omb1, gi));
  const float vi = __fmaf_rn(beta2, v[i], __fmul_rn(__fmul_rn(omb2, gi), gi));
  m[i] = mi;
  v[i] = vi;
  p[i] = __fsub_rn(p[i], __fdiv_rn(__fmul_rn(step_size, mi), __fmaf_rn(__fsqrt_rn(vi), inv_sqrt_bc2, eps)));
}

__global__ void __launch_bounds__(256)
adam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                 long long n, float step_size, float beta1, float beta2, float omb1, float omb2, float inv_sqrt_bc2,
                 float eps, int zero_grad) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    adam_update(p, m, v, i, g[i], step_size, beta1, beta2, omb1, omb2, inv_sqrt_bc2, eps);
    if (zero_grad) g[i] = 0.f;
  }
}

// torch.optim.RAdam (the optimiser of fruit_nerf_method_big / _huge, fruit_nerf_config.py:101-117,151-167): the Adam
// moments, then either the variance-rectified step m_hat * lr * rect * sqrt(bc2) / (sqrt(v) + eps) or, while the variance
// estimate is not yet tractable (rho_t <= 5, the first steps), the plain momentum step m_hat * lr.
__global__ void __launch_bounds__(256)
radam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                  long long n, float step_size, float beta1, float beta2, float omb1, float omb2, float eps,
                  int rectified, int zero_grad) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = beta1 * m[i] + omb1 * gi;
    const float vi = beta2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= rectified ? step_size * mi / (sqrtf(vi) + eps) : step_size * mi;
    if (zero_grad) g[i] = 0.f;
  }
}

// the same update with its per-step scalars in device memory (hyper = {step_size, beta1, beta2, 1 - beta1, 1 - beta2,
// 1 / sqrt(bias correction 2), eps}): a captured HIP graph of the training iteration replays with the schedule's new values
__global__ void __launch_bounds__(256)
adam_step_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                     const float* __restrict__ hyper, int zero_grad) {
  const float step_size = hyper[0], beta1 = hyper[1], beta2 = hyper[2], omb1 = hyper[3], omb2 = hyper[4],
              inv_sqrt_bc2 = hyper[5], eps = hyper[6];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    adam_update(p, m, v, i, g[i], step_size, beta1, beta2, omb1, omb2, inv_sqrt_bc2, eps);
    if (zero_grad) g[i] = 0.f;
  }
}

// every optimiser group of a flat parameter buffer in ONE launch: group k owns [bounds[k], bounds[k + 1]); hyper row k as in
// adam_step_dev_kernel, hyper[k][7] != 0 marks a group WITHOUT a step this iteration (frozen, or the proposal networks between
// their updates): its gradients are only zeroed
constexpr int ADAM_MAX_GROUPS = 8;
struct AdamGroups {
  long long bounds[ADAM_MAX_GROUPS + 1];
  int num;
};
__global__ void __launch_bounds__(256)
adam_step_groups_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, AdamGroups G,
                        const float* __restrict__ hyper) {
  const long long n = G.bounds[G.num];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < ADAM_MAX_GROUPS; ++j) k += (j < G.num && i >= G.bounds[j]) ? 1 : 0;
    const float* h = hyper + 8 * k;
    if (h[7] == 0.f) adam_update(p, m, v, i, g[i], h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    g[i] = 0.f;
  }
}

}  // namespace cn

extern "C" int cn_train_render_backward(const float* starts, const float* ends, const float* density,
                                        const float* rgb, const float* semantics, const float* image,
                                        const float* fruit_mask, int64_t num_rays, int32_t num_samples,
                                        float semantic_loss_weight, float* out_rgb, float* out_semantics,
                                        float* out_accumulation, float* out_weights, float* d_density, float* d_rgb,
                                        float* d_semantics, float* loss_sums, const float* spacing_bins,
                                        cn_stream_t stream) {
  CN_REQUIRE(starts && ends && density && rgb && semantics && image && fruit_mask && d_density && d_rgb &&
                 d_semantics && loss_sums,
             CN_ERR_INVALID, "cn_train_render_backward: null argument");
  CN_REQUIRE(num_samples >= 1 && num_samples <= cn::TRAIN_MAX_S, CN_ERR_UNSUPPORTED,
             "cn_train_render_backward: %d samples per ray (max %d)", num_samples, cn::TRAIN_MAX_S);
  if (num_rays <= 0) return CN_OK;
  size_t lds = (size_t)4 * 2 * num_samples * sizeof(float);
  hipLaunchKernelGGL(cn::train_render_backward_kernel, dim3(cn::grid_for(num_rays, 4, 4096)), dim3(256), lds,
                     cn::as_stream(stream), starts, ends, density, rgb, semantics, image, fruit_mask,
                     (long long)num_rays, num_samples, semantic_loss_weight, out_rgb, out_semantics, out_accumulation,
                     out_weights, d_density, d_rgb, d_semantics, loss_sums, spacing_bins);
  if (int rc = cn::check_launch("cn_train_render_backward")) return rc;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" int cn_interlevel_backward(const float* final_spacing_bins, const float* final_weights,
                                      const float* prop_spacing_bins, const float* prop_starts, const float* prop_ends,
                                      const float* prop_density, int64_t num_rays, int32_t s_final, int32_t s_prop,
                                      float loss_mult, float* d_prop_density, float* loss_sum, cn_stream_t stream) {
  CN_REQUIRE(final_spacing_bins && final_weights && prop_spacing_bins && prop_starts && prop_ends && prop_density &&
                 d_prop_density && loss_sum,
             CN_ERR_INVALID, "cn_interlevel_backward: null argument");
  CN_REQUIRE(s_final >= 1 && s_prop >= 1 && s_prop <= cn::TRAIN_MAX_S, CN_ERR_UNSUPPORTED,
             "cn_interlevel_backward: sample counts %d / %d", s_final, s_prop);
  if (num_rays <= 0) return CN_OK;
  size_t lds = (size_t)4 * (5 * s_prop + 3) * sizeof(float);
  float mult = loss_mult / ((float)num_rays * (float)s_final);
  hipLaunchKernelGGL(cn::interlevel_backward_kernel, dim3(cn::grid_for(num_rays, 4, 4096)), dim3(256), lds,
                     cn::as_stream(stream), final_spacing_bins, final_weights, prop_spacing_bins, prop_starts, prop_ends,
                     prop_density, (long long)num_rays, s_final, s_prop, mult, d_prop_density, loss_sum);
  if (int rc = cn::check_launch("cn_interlevel_backward")) return rc;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" int cn_interlevel_backward_levels(const float* final_spacing_bins, const float* final_weights,
                                             const cn_interlevel_level* levels, int32_t num_levels, int64_t num_rays,
                                             int32_t s_final, float loss_mult, float* loss_sum, cn_stream_t stream) {
  CN_REQUIRE(final_spacing_bins && final_weights && levels && loss_sum, CN_ERR_INVALID, "cn_interlevel_backward_levels: null argument");
  CN_REQUIRE(num_levels >= 1 && num_levels <= cn::INTERLEVEL_MAX_LEVELS, CN_ERR_UNSUPPORTED,
             "cn_interlevel_backward_levels: %d levels (max %d)", num_levels, cn::INTERLEVEL_MAX_LEVELS);
  CN_REQUIRE(s_final >= 1, CN_ERR_UNSUPPORTED, "cn_interlevel_backward_levels: %d final samples", s_final);
  cn::InterlevelLevels L{};
  int smax = 0;
  for (int k = 0; k < num_levels; ++k) {
    const cn_interlevel_level& v = levels[k];
    CN_REQUIRE(v.spacing_bins && v.starts && v.ends && v.density && v.d_density, CN_ERR_INVALID,
               "cn_interlevel_backward_levels: null buffer in level %d", k);
    CN_REQUIRE(v.num_samples >= 1 && v.num_samples <= cn::TRAIN_MAX_S, CN_ERR_UNSUPPORTED,
               "cn_interlevel_backward_levels: level %d has %d samples per ray (max %d)", k, v.num_samples, cn::TRAIN_MAX_S);
    L.cp_bins[k] = v.spacing_bins;
    L.starts[k] = v.starts;
    L.ends[k] = v.ends;
    L.density[k] = v.density;
    L.d_density[k] = v.d_density;
    L.Sp[k] = v.num_samples;
    smax = std::max(smax, (int)v.num_samples);
  }
  for (int k = num_levels; k < cn::INTERLEVEL_MAX_LEVELS; ++k) {  // (never selected: blockIdx.y < num_levels)
    L.cp_bins[k] = L.cp_bins[0];
    L.starts[k] = L.starts[0];
    L.ends[k] = L.ends[0];
    L.density[k] = L.density[0];
    L.d_density[k] = L.d_density[0];
    L.Sp[k] = L.Sp[0];
  }
  if (num_rays <= 0) return CN_OK;
  const size_t lds = (size_t)4 * (5 * smax + 3) * sizeof(float);
  const float mult = loss_mult / ((float)num_rays * (float)s_final);
  hipLaunchKernelGGL(cn::interlevel_backward_levels_kernel, dim3(cn::grid_for(num_rays, 4, 4096), num_levels), dim3(256), lds,
                     cn::as_stream(stream), final_spacing_bins, final_weights, L, (long long)num_rays, s_final, mult, loss_sum);
  if (int rc = cn::check_launch("cn_interlevel_backward_levels")) return rc;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" int cn_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                            double lr, double beta1, double beta2, double eps, int32_t zero_grad, cn_stream_t stream) {
  CN_REQUIRE(param && grad && exp_avg && exp_avg_sq, CN_ERR_INVALID, "cn_adam_step: null argument");
  CN_REQUIRE(step >= 1, CN_ERR_INVALID, "cn_adam_step: step is 1-based");
  if (n <= 0) return CN_OK;
  // hyper-parameters arrive as doubles (Python floats) so that 1-beta rounds to fp32 exactly as torch's does
  double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(cn::adam_step_kernel, dim3(cn::grid_for(n, 256, 4096)), dim3(256), 0, cn::as_stream(stream), param,
                     grad, exp_avg, exp_avg_sq, (long long)n, (float)(lr / bc1), (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), (float)(1.0 / sqrt(bc2)), (float)eps, zero_grad);
  return cn::check_launch("cn_adam_step");
}

extern "C" int cn_adam_hyper(int32_t step, double lr, double beta1, double beta2, double eps, float* hyper_host) {
  CN_REQUIRE(hyper_host && step >= 1, CN_ERR_INVALID, "cn_adam_hyper: bad argument (step is 1-based)");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hyper_host[0] = (float)(lr / bc1);
  hyper_host[1] = (float)beta1;
  hyper_host[2] = (float)beta2;
  hyper_host[3] = (float)(1.0 - beta1);
  hyper_host[4] = (float)(1.0 - beta2);
  hyper_host[5] = (float)(1.0 / sqrt(bc2));
  hyper_host[6] = (float)eps;
  hyper_host[7] = 0.f;
  return CN_OK;
}

extern "C" int cn_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper,
                                int32_t zero_grad, cn_stream_t stream) {
  CN_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper, CN_ERR_INVALID, "cn_adam_step_dev: null argument");
  if (n <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::adam_step_dev_kernel, dim3(cn::grid_for(n, 256, 4096)), dim3(256), 0, cn::as_stream(stream), param,
                     grad, exp_avg, exp_avg_sq, (long long)n, hyper, zero_grad);
  return cn::check_launch("cn_adam_step_dev");
}

extern "C" int cn_adam_step_groups_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* bounds_host,
                                       int32_t num_groups, const float* hyper, cn_stream_t stream) {
  CN_REQUIRE(param && grad && exp_avg && exp_avg_sq && bounds_host && hyper, CN_ERR_INVALID,
             "cn_adam_step_groups_dev: null argument");
  CN_REQUIRE(num_groups >= 1 && num_groups <= cn::ADAM_MAX_GROUPS, CN_ERR_UNSUPPORTED,
             "cn_adam_step_groups_dev: %d groups (max %d)", num_groups, cn::ADAM_MAX_GROUPS);
  cn::AdamGroups G{};
  G.num = num_groups;
  for (int k = 0; k <= num_groups; ++k) {
    CN_REQUIRE(k == 0 ? bounds_host[0] == 0 : bounds_host[k] >= bounds_host[k - 1], CN_ERR_INVALID,
               "cn_adam_step_groups_dev: group bounds must start at 0 and ascend");
    G.bounds[k] = bounds_host[k];
  }
  const long long n = G.bounds[num_groups];
  if (n <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::adam_step_groups_kernel, dim3(cn::grid_for(n, 256, 4096)), dim3(256), 0, cn::as_stream(stream), param,
                     grad, exp_avg, exp_avg_sq, G, hyper);
  return cn::check_launch("cn_adam_step_groups_dev");
}

extern "C" int cn_radam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                             double lr, double beta1, double beta2, double eps, int32_t zero_grad, cn_stream_t stream) {
  CN_REQUIRE(param && grad && exp_avg && exp_avg_sq, CN_ERR_INVALID, "cn_radam_step: null argument");
  CN_REQUIRE(step >= 1, CN_ERR_INVALID, "cn_radam_step: step is 1-based");
  if (n <= 0) return CN_OK;
  const double b2t = pow(beta2, (double)step), bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - b2t;
  const double rho_inf = 2.0 / (1.0 - beta2) - 1.0, rho_t = rho_inf - 2.0 * step * b2t / bc2;
  const int rectified = rho_t > 5.0;
  double step_size = lr / bc1;
  if (rectified)
    step_size *= sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) * sqrt(bc2);
  hipLaunchKernelGGL(cn::radam_step_kernel, dim3(cn::grid_for(n, 256, 4096)), dim3(256), 0, cn::as_stream(stream), param,
                     grad, exp_avg, exp_avg_sq, (long long)n, (float)step_size, (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, rectified, zero_grad);
  return cn::check_launch("cn_radam_step");
}

extern "C" int cn_distortion_metric(const float* spacing_bins, const float* weights, int64_t num_rays,
                                    int32_t num_samples, float* sum_out, cn_stream_t stream) {
  CN_REQUIRE(spacing_bins && weights && sum_out, CN_ERR_INVALID, "cn_distortion_metric: null argument");
  CN_REQUIRE(num_samples >= 1 && num_samples <= cn::TRAIN_MAX_S, CN_ERR_UNSUPPORTED,
             "cn_distortion_metric: %d samples per ray (max %d)", num_samples, cn::TRAIN_MAX_S);
  if (num_rays <= 0) return CN_OK;
  size_t lds = (size_t)4 * 2 * num_samples * sizeof(float);
  hipLaunchKernelGGL(cn::distortion_kernel, dim3(cn::grid_for(num_rays, 4, 4096)), dim3(256), lds, cn::as_stream(stream),
                     spacing_bins, weights, (long long)num_rays, num_samples, sum_out);
  if (int rc = cn::check_launch("cn_distortion_metric")) return rc;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" int cn_ray_backward(const float* d_positions, const float* d_dir_samples, const float* starts,
                               const float* ends, int64_t num_rays, int32_t num_samples, float* d_origins,
                               float* d_directions, cn_stream_t stream) {
  CN_REQUIRE(d_positions && starts && ends && d_origins && d_directions, CN_ERR_INVALID,
             "cn_ray_backward: null argument");
  CN_REQUIRE(num_samples >= 1, CN_ERR_INVALID, "cn_ray_backward: num_samples must be >= 1");
  if (num_rays <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::ray_backward_kernel, dim3(cn::grid_for(num_rays, 4, 8192)), dim3(256), 0,
                     cn::as_stream(stream), d_positions, d_dir_samples, starts, ends, (long long)num_rays, num_samples,
                     d_origins, d_directions);
  return cn::check_launch("cn_ray_backward");
}

extern "C" int cn_pose_adjustment_backward(const float* pose_adjustment, const int64_t* camera_indices,
                                           const float* directions_raw, const float* d_origins,
                                           const float* d_directions, int64_t num_rays, int32_t num_cameras,
                                           float* grad_pose, cn_stream_t stream) {
  CN_REQUIRE(pose_adjustment && camera_indices && directions_raw && d_origins && d_directions && grad_pose,
             CN_ERR_INVALID, "cn_pose_adjustment_backward: null argument");
  if (num_rays <= 0) return CN_OK;
  // (the deterministic test build takes the second form: the LDS sums of the first are float atomics of four waves)
  if (!CN_DETERMINISTIC_SCATTER && num_cameras > 0 && num_cameras <= 2048)  // (48 KB of LDS at most; 0 = unknown: one atomic per ray and entry)
    hipLaunchKernelGGL(cn::pose_backward_kernel<true>, dim3(cn::grid_for(num_rays, 1024, 64)), dim3(256),
                       (size_t)6 * num_cameras * sizeof(float), cn::as_stream(stream), pose_adjustment, camera_indices,
                       directions_raw, d_origins, d_directions, (long long)num_rays, num_cameras, grad_pose);
  else
    hipLaunchKernelGGL(cn::pose_backward_kernel<false>, dim3(cn::grid_for(num_rays, 256, 4096)), dim3(256), 0,
                       cn::as_stream(stream), pose_adjustment, camera_indices, directions_raw, d_origins, d_directions,
                       (long long)num_rays, 0, grad_pose);
  if (int rc = cn::check_launch("cn_pose_adjustment_backward")) return rc;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

extern "C" int cn_train_epilogue(const float* loss_sums, int32_t num_sums, int64_t num_rays, int32_t num_samples,
                                 float semantic_loss_weight, float interlevel_loss_mult, const float* pose_adjustment,
                                 int32_t num_cameras, float* out, cn_stream_t stream) {
  CN_REQUIRE(loss_sums && out && num_rays > 0 && num_samples > 0 && (num_sums == 4 || num_sums == 5), CN_ERR_INVALID,
             "cn_train_epilogue: bad argument (loss_sums holds 4 sums, or 5 with the distortion sum)");
  const double r = (double)num_rays;
  hipLaunchKernelGGL(cn::train_epilogue_kernel, dim3(1), dim3(64), 0, cn::as_stream(stream), loss_sums, 1.0 / (3.0 * r),
                     (double)semantic_loss_weight / r, (double)interlevel_loss_mult / (r * (double)num_samples), 1.0 / r,
                     num_sums == 5 ? 1 : 0, pose_adjustment, pose_adjustment ? num_cameras : 0, out);
  return cn::check_launch("cn_train_epilogue");
}

extern "C" int cn_pose_regularizer(const float* pose_adjustment, int32_t num_cameras, float trans_l2_penalty,
                                   float rot_l2_penalty, float* grad_pose, float* loss_out, cn_stream_t stream) {
  CN_REQUIRE(pose_adjustment && loss_out, CN_ERR_INVALID, "cn_pose_regularizer: null argument");
  if (num_cameras <= 0) return CN_OK;
  hipLaunchKernelGGL(cn::pose_regularizer_kernel, dim3(cn::grid_for(num_cameras, 256, 1024)), dim3(256), 0,
                     cn::as_stream(stream), pose_adjustment, num_cameras, trans_l2_penalty, rot_l2_penalty, grad_pose,
                     loss_out);
  if (int rc = cn::check_launch("cn_pose_regularizer")) return rc;
  CN_DET_FLUSH(cn::as_stream(stream));
  return CN_OK;
}

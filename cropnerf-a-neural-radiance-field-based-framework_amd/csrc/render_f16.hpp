// render_f16_kernel -- the single-wave renderer in the reference's OWN arithmetic class (cn_render_opts.matrix_precision =
// CN_MATRIX_F16): tiny-cuda-nn's FullyFusedMLP under mixed precision (fruit_nerf/fruit_field.py:95,125-167 build every
// module with implementation="tcnn"; fruit_nerf_config.py:35 mixed_precision=True).  Included by render_fused.hip.
//
// What it is for.  The DENSITY-ONLY pass of the fp16 mode (get_density_for_camera_ray_bundle, the occlusion pass of the
// projection: render_split_kernel has no such variant), and an A/B alternative for the composited and per-sample renders
// (CN_F16_KERNEL=own).  It was built as THE fp16 kernel on this reasoning: with fp16 operands the MLP chain of 32 samples is 44
// v_mfma_f32_16x16x32_f16 instead of 288 fp32 MFMAs, the kernel is its hash-grid gathers, and the producer/consumer kernel's
// barrier per half-step keeps its gather waves in lock-step -- so let every wave own a ray end to end, 16-20 independent waves
// per CU (28.8 KB LDS image: five workgroups fit), one wave's memory wait another's arithmetic.  Measured at C2 on a tcnn fp16
// table: 3.83 ms against the split kernel's 2.15 (2.87 against 1.77 once both had the x-pair gathers; 1.39 now).  What the
// gathers cost is L1 line lookups and L1 misses, not exposed latency: twice the rays in flight per CU, at
// unrelated depths, evict each other's lines from the 32 KB L1, while the split kernel's eight rays -- neighbouring pixels at
// the SAME depth -- share them.  Lock-step is a feature there.
//
// Arithmetic (the parity bar is oracle/tcnn.py with tcnn_half_activations=True, tests/test_gpu_f16.py):
//   * weights rounded to fp16 in prep_kernel (a no-op for an imported tcnn checkpoint, whose parameters are fp16 values);
//   * half table: the trilinear blend runs on packed fp16 pairs (hash_level_pk_*: v_pk_fma_f16, 14 instructions for both
//     features, x-pair gathers), as tcnn's kernel_grid accumulates in the parameter type; float table: fp32 blend, result
//     rounded to fp16;
//   * every layer input rounded to fp16 (v_cvt_pk_f16_f32, round to nearest even), products summed in fp32 by the MFMA
//     (tcnn accumulates in fp16: this is at least as precise), biases / heads / sigmoid / compositing in fp32.
// Ray -> wave scheduling (XCD column stripes), compositing and early termination are render_fused_kernel's.
#pragma once

namespace cn {

#ifndef CN_F16_WAVES_PER_SIMD
#define CN_F16_WAVES_PER_SIMD 4
#endif
#ifndef CN_F16_LEVELS_IN_FLIGHT
#define CN_F16_LEVELS_IN_FLIGHT 2
#endif

// LDS image of a workgroup: the 22 fp16 A-operand blocks, then the blob from OFF_B0 on (biases, folded heads, SH columns,
// level records) -- 28.8 KB instead of the 43 KB of the fp32 image, so that five 4-wave workgroups fit a CU.
constexpr int F16_IMG_FLOATS = BF16_BLOCKS * 256;
constexpr int F16_TAIL_FLOATS = BLOB_FLOATS - OFF_B0;
constexpr int F16_LDS_FLOATS = F16_IMG_FLOATS + F16_TAIL_FLOATS + FUSED_WAVES * WAVE_SCRATCH;
static_assert(F16_IMG_FLOATS % 4 == 0 && F16_TAIL_FLOATS % 4 == 0 && OFF_B0 % 4 == 0, "copied as float4");
#define CN_F16_T(OFF) (F16_IMG_FLOATS + (OFF) - OFF_B0)  // LDS float index of blob offset OFF (>= OFF_B0)

template <bool PER_SAMPLE, bool DENSITY_ONLY, bool HALF, bool GENERIC>
__global__ void __launch_bounds__(FUSED_THREADS, CN_F16_WAVES_PER_SIMD) render_f16_kernel(FusedArgs A) {
  __shared__ __align__(16) float lds[F16_LDS_FLOATS];
  {
    const float4* src = reinterpret_cast<const float4*>(A.blob);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < F16_IMG_FLOATS / 4; i += FUSED_THREADS) dst[i] = src[i];
    for (int i = threadIdx.x; i < F16_TAIL_FLOATS / 4; i += FUSED_THREADS) dst[F16_IMG_FLOATS / 4 + i] = src[OFF_B0 / 4 + i];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int g = lane >> 4, j = lane & 15;
  float* scratch = lds + F16_IMG_FLOATS + F16_TAIL_FLOATS + wave * WAVE_SCRATCH;  // [0,64): colour bias, [64,129): bin edges
  float* tbuf = scratch + 64;
  const int S = A.S;
  auto blkh = [&](int b) { return reinterpret_cast<const f16x8*>(lds + b * 256)[lane]; };

  // work items: rays (composited render) or (ray, 64-sample chunk) pairs (per-sample outputs have no dependency along a
  // ray, so the exporters' 512-ray x 3000-sample calls fill the device)
  const int nchunks = (S + 63) >> 6;
  const long long nwork = PER_SAMPLE ? A.num_rays * nchunks : A.num_rays;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const long long stride = (long long)(gridDim.x >> 3) * FUSED_WAVES;
  const bool striped = !PER_SAMPLE && A.image_width > 0;
  const long long per_xcd = (nwork + 7) >> 3;
  const int nstripe = 8 * A.stripes_per_xcd;
  const int cw = striped ? (A.image_width + nstripe - 1) / nstripe : 0;
  const long long first_row = striped ? A.pixel_start / A.image_width : 0;
  const long long last_row = striped ? (A.pixel_start + A.num_rays - 1) / A.image_width : 0;
  const long long rows = last_row - first_row + 1;
  const long long items = striped ? rows * cw * A.stripes_per_xcd : min(per_xcd, max(nwork - xcd * per_xcd, 0LL));

  for (long long q = slot * FUSED_WAVES + wave; q < items; q += stride) {
    long long rr;
    if (striped) {
      const long long sq = q / (rows * cw);
      const long long qq = q - sq * rows * cw;
      const long long vrow = qq / cw;
      const int col = (int)(sq * 8 + xcd) * cw + (int)(qq - vrow * cw);
      rr = (first_row + vrow) * A.image_width + col - A.pixel_start;
      if (col >= A.image_width || rr < 0 || rr >= A.num_rays) continue;  // wave-uniform
    } else {
      rr = xcd * per_xcd + q;
    }
    int chunk_first = 0, chunk_end = nchunks;
    if (PER_SAMPLE) {
      chunk_first = (int)(rr % nchunks);
      chunk_end = chunk_first + 1;
      rr /= nchunks;
    }
    const long long r = __builtin_amdgcn_readfirstlane((int)rr);
    const float ox = A.origins[3 * r], oy = A.origins[3 * r + 1], oz = A.origins[3 * r + 2];
    const float dx = A.directions[3 * r], dy = A.directions[3 * r + 1], dz = A.directions[3 * r + 2];
    const float sn = spacing_fn(A.spacing, A.nears[r]), sf = spacing_fn(A.spacing, A.fars[r]);
    const float* bins = A.bins ? A.bins + r * (long long)(S + 1) : nullptr;
    auto edge = [&](int i) -> float {
      i = min(i, S);
      return bins ? bins[i] : spacing_to_euclid(A.spacing, linspace01(i, S + 1), sn, sf);
    };

    if (!DENSITY_ONLY) {
      // per-ray colour bias: bc0 + Wc0[:, sh] . SH(d) + Wc0[:, app] . app, on fp16-rounded inputs (lane n = neuron n)
      float sx = dx, sy = dy, sz = dz;
      if (!A.sh_unit) {
        sx = (dx + 1.f) / 2.f;
        sy = (dy + 1.f) / 2.f;
        sz = (dz + 1.f) / 2.f;
      }
      float sh[16];
      sh_deg4(sx, sy, sz, sh);
#pragma unroll
      for (int k = 0; k < 16; ++k) sh[k] = (float)(_Float16)sh[k];
      const long long row = A.app_per_camera ? A.cam_idx[r] : 0;
      float bias = A.app_bias[row * 64 + lane];
      const f32x4* wsh = reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_WSH) + lane * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 w = wsh[k];
        bias = fmaf(w.x, sh[4 * k + 0], bias);
        bias = fmaf(w.y, sh[4 * k + 1], bias);
        bias = fmaf(w.z, sh[4 * k + 2], bias);
        bias = fmaf(w.w, sh[4 * k + 3], bias);
      }
      __builtin_amdgcn_wave_barrier();
      scratch[lane] = bias;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    CompositeState st;
    for (int ch = chunk_first; ch < chunk_end; ++ch) {
      const int c0 = ch * 64;
      const float e_lo = edge(c0 + lane);
      const float e_top = edge(c0 + 64);
      __builtin_amdgcn_wave_barrier();
      tbuf[lane] = e_lo;
      if (lane == 0) tbuf[64] = e_top;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

      float my_dlogit = 0.f, my_sel = 0.f, my_sem = 0.f, my_r = 0.f, my_g = 0.f, my_b = 0.f;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        float px[2], py[2], pz[2];
        bool sel[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int k = 32 * half + 16 * c + j;
          const float mid = (tbuf[k] + tbuf[k + 1]) / 2.f;
          px[c] = ox + dx * mid;
          py[c] = oy + dy * mid;
          pz[c] = oz + dz * mid;
          sel[c] = normalize_position(A.scene, px[c], py[c], pz[c]);
        }
        // ---- hash grid: levels 4g..4g+3 of the two samples, as the packed fp16 pairs the first MFMA consumes ----------
        u32x4v featp[2];
        {
          const f32x4 lvl_scale = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_SCALE) + 4 * g);
          const float pos_off = GENERIC ? A.grid.pos_offset : 0.f;
          if constexpr (HALF && !CN_ABLATE_GATHER) {
            // pair gathers: the eight (level, sample) units in program order, unit n + 1's loads issued before unit n is blended
            PkLoads cur = hash_level_pk_issue<GENERIC>(
                A.grid.table, lane_level_rec<GENERIC>(lds + CN_F16_T(OFF_LVL), A.grid, 4 * g, lvl_scale[0]), pos_off, px[0], py[0], pz[0]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              PkLoads nxt;
              if (u < 7) {
                const int qn = (u + 1) >> 1, cn_ = (u + 1) & 1;
                nxt = hash_level_pk_issue<GENERIC>(
                    A.grid.table, lane_level_rec<GENERIC>(lds + CN_F16_T(OFF_LVL), A.grid, 4 * g + qn, lvl_scale[qn]), pos_off, px[cn_],
                    py[cn_], pz[cn_]);
              }
              featp[u & 1][u >> 1] = pk_pin(hash_level_pk_blend(cur));
              if (u < 7) cur = nxt;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const Lvl lv = lane_level_rec<GENERIC>(lds + CN_F16_T(OFF_LVL), A.grid, 4 * g + q, lvl_scale[q]);
#pragma unroll
              for (int c = 0; c < 2; ++c) {
#if CN_ABLATE_GATHER  // timing-only build: no table reads
                featp[c][q] = __builtin_bit_cast(unsigned, px[c] * lv.scale + py[c]) ^ __builtin_bit_cast(unsigned, pz[c]);
#else
                const float2 f = hash_level_sc<HALF, GENERIC>(A.grid.table, lv, pos_off, px[c], py[c], pz[c]);
                const f16x2 hp = {(_Float16)f.x, (_Float16)f.y};
                featp[c][q] = __builtin_bit_cast(unsigned, hp);
#endif
              }
              if ((q + 1) % CN_F16_LEVELS_IN_FLIGHT == 0) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- base MLP: 32 -> 64 ReLU -> 16 (neuron 0 = density logit, 1..15 = geo features) -----------------------------
        f16x8 hh[2][2];  // [K block][column tile]: the hidden layer as the next layer's B operands
        {
          f32x4 h[4][2];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_B0) + 16 * mt + 4 * g);
            const f16x8 a = blkh(mt);
#pragma unroll
            for (int c = 0; c < 2; ++c) h[mt][c] = relu4(mfma_f16(a, __builtin_bit_cast(f16x8, featp[c]), b));
          }
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < 2; ++c) hh[kb][c] = cvt_f16x8(h[2 * kb][c], h[2 * kb + 1][c]);
        }
        f32x4 o16[2];
        {
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_B1) + 4 * g);
          o16[0] = b1;
          o16[1] = b1;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const f16x8 a = blkh(4 + kb);
#pragma unroll
            for (int c = 0; c < 2; ++c) o16[c] = mfma_f16(a, hh[kb][c], o16[c]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool mine = (g >> 1) == half;  // this half holds lane l's own sample in column tile g & 1
        const bool odd = (g & 1) != 0;
        {
          const float d0 = row0_broadcast(o16[0].x), d1 = row0_broadcast(o16[1].x);
          my_dlogit = mine ? (odd ? d1 : d0) : my_dlogit;
          my_sel = mine ? ((odd ? sel[1] : sel[0]) ? 1.f : 0.f) : my_sel;
        }
        if (!DENSITY_ONLY) {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          f16x8 oh[2];  // the 16 base outputs as a K block of 32 (upper half zero; the weight image zeroes neuron 0)
#pragma unroll
          for (int c = 0; c < 2; ++c) oh[c] = cvt_f16x8(o16[c], zero4);
          // ---- semantics: relu(Ws0 geo + bs0) . (Wh Ws1) + folded bias ----------------------------------------------------
          float sem_part[2] = {0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_BS0) + 16 * mt + 4 * g);
            const f32x4 wf = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_WF) + 16 * mt + 4 * g);
            const f16x8 a = blkh(6 + mt);
#pragma unroll
            for (int c = 0; c < 2; ++c) sem_part[c] = dot4(wf, relu4(mfma_f16(a, oh[c], b)), sem_part[c]);
          }
          // ---- colour layer 0: geo columns on the MFMA, SH + appearance columns pre-summed in the ray bias -----------------
          f16x8 ch[2][2];
          {
            f32x4 c1[4][2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              const f32x4 cb = *reinterpret_cast<const f32x4*>(scratch + 16 * mt + 4 * g);
              const f16x8 a = blkh(10 + mt);
#pragma unroll
              for (int c = 0; c < 2; ++c) c1[mt][c] = relu4(mfma_f16(a, oh[c], cb));
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int c = 0; c < 2; ++c) ch[kb][c] = cvt_f16x8(c1[2 * kb][c], c1[2 * kb + 1][c]);
          }
          __builtin_amdgcn_sched_barrier(0);
          // ---- colour layer 1 (64 -> 64, ReLU) with the 64 -> 3 head folded into the row-tile loop ---------------------------
          float rgb_part[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_BC1) + 16 * mt + 4 * g);
            f32x4 acc[2] = {b, b};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              const f16x8 a = blkh(14 + 2 * mt + kb);
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = mfma_f16(a, ch[kb][c], acc[c]);
            }
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_WRGB) + 0 * 64 + 16 * mt + 4 * g);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_WRGB) + 1 * 64 + 16 * mt + 4 * g);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds + CN_F16_T(OFF_WRGB) + 2 * 64 + 16 * mt + 4 * g);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f32x4 v = relu4(acc[c]);
              rgb_part[c][0] = dot4(w0, v, rgb_part[c][0]);
              rgb_part[c][1] = dot4(w1, v, rgb_part[c][1]);
              rgb_part[c][2] = dot4(w2, v, rgb_part[c][2]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          const float s0 = group_sum(sem_part[0]), s1 = group_sum(sem_part[1]);
          my_sem = mine ? (odd ? s1 : s0) : my_sem;
          const float r0 = group_sum(rgb_part[0][0]), r1 = group_sum(rgb_part[1][0]);
          my_r = mine ? (odd ? r1 : r0) : my_r;
          const float g0 = group_sum(rgb_part[0][1]), g1 = group_sum(rgb_part[1][1]);
          my_g = mine ? (odd ? g1 : g0) : my_g;
          const float b0 = group_sum(rgb_part[0][2]), b1 = group_sum(rgb_part[1][2]);
          my_b = mine ? (odd ? b1 : b0) : my_b;
        }
      }
      // ---- lane l now holds sample c0 + l -----------------------------------------------------------------------------------
      const float density = expf(my_dlogit) * my_sel;
      float sem = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
      if (!DENSITY_ONLY) {
        sem = my_sem + lds[CN_F16_T(OFF_MISC) + 0];
        cr = sigmoidf(my_r + lds[CN_F16_T(OFF_MISC) + 1]);
        cg = sigmoidf(my_g + lds[CN_F16_T(OFF_MISC) + 2]);
        cb = sigmoidf(my_b + lds[CN_F16_T(OFF_MISC) + 3]);
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
        const float w = composite_chunk(st, valid, i == S - 1, e1 - e0, density, mid, cr, cg, cb, sem, A.eval_clamp != 0);
        if (A.out_w && valid) A.out_w[r * (long long)S + i] = w;
        // optional early ray termination, as render_fused_kernel (off by default)
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
      const CompositeOut o = composite_finish(st, A.bg_mode, A.bg[0], A.bg[1], A.bg[2], A.eval_clamp != 0);
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
            const float l = semantics_label(o.sem);
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

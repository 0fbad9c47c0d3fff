// Producer / consumer form of render_fused_kernel<false,false> (included by render_fused.hip).
//
// The fused kernel is bound by the rate at which a CU's texture-address unit takes divergent 8-byte gathers (~1.25
// lanes per clock), with the matrix work (~80 % of that time) hidden under it only partly: a wave that is inside its
// MFMA chain issues no gathers, so the TA idles whenever too many of the 12 resident waves are in that phase.  Here the
// two halves of the per-sample work run in DIFFERENT waves of one 1024-thread workgroup (one per CU):
//   * 8 gather waves: positions -> hash -> 128 corner gathers per sample -> trilinear blend -> 32 features, written to LDS
//     in the lane layout the first MFMA wants (a register hand-off: 4 x 16-byte stores per lane);
//   * 8 matrix waves: features from LDS -> the MFMA chain of the field -> heads -> wave-scan compositing.
// Gather wave p feeds matrix wave p one 32-sample half-step ahead through a two-slot ring; one workgroup barrier per
// half-step is the only synchronisation.  Gather waves never wait for matrix work, so the TA always has requests queued,
// and the matrix waves (2 per SIMD) never wait for memory.  Ray -> pair scheduling, arithmetic and outputs are exactly
// those of render_fused_kernel (tests compare the two; the binaries differ by an ulp where hipcc contracts
// multiply-adds differently).  Measured at C2: 3.24 -> 2.56 ms per 65 536-ray batch.
#pragma once

namespace cn {

// levels whose gathers a gather wave keeps in flight per lane (16 loads each; measured at C2: 1 -> 4.83, 2 -> 4.85,
// 4 -> 4.92 Gsamples/s -- a gather wave has the registers to spare)
#ifndef CN_SPLIT_LEVELS_IN_FLIGHT
#define CN_SPLIT_LEVELS_IN_FLIGHT 4
#endif
// CN_SPLIT_G gather waves, each feeding CN_SPLIT_MPG matrix waves ("pairs" below = matrix waves = rays in flight).
// Measured at C2 (Gsamples/s), then removed from the source again (the commits are in the history):
//   G x MPG = 4x3 / 5x2 / 4x2 / 6x1 instead of 8x1                                                   3.49 / 3.82 / 4.43 / 4.17 vs 4.93
//   per-pair full/empty flags in LDS instead of the workgroup barrier                              4.31 vs 4.89
//   one loop per role instead of one loop with a role branch                                        4.65 vs 4.93
//   the gather waves also run the base MLP (96 of the 288 MFMAs) and hand over its 16 outputs        4.29 vs 4.87
//   one of the two matrix waves of a SIMD defers its compositing to its next half-step               4.88 vs 4.89
//   round 3: pairs that advance independently (the gather side owns a pair's progress and labels every half-step in the
//   ring with its schedule slot and index, the matrix wave follows the labels, the workgroup runs until every pair reports
//   its slot list exhausted), so that a pair whose ray terminated early takes its next slot instead of idling: results
//   bit-identical, but 2.69 vs 2.60 ms on the headline, 1.72 vs 1.60 ms in fp16 mode (labels and flags through LDS, ten more
//   registers), and NO gain on the opaque C2 batch of tools/early_stop_probe.py (1.126 ms either way): there every ray
//   stops after its first chunk, and what the split kernel pays over render_fused_kernel (0.87 ms) is the half-step the
//   gather wave is always ahead of its matrix wave -- produced for a ray that has just been terminated -- not lock-step
//   idling.  Interleaving two rays per pair at chunk granularity would hide it (the decision on ray A falls while the
//   gather wave works on ray B) at the price of two compositing states per matrix wave; not built.
// (with fewer gather waves the gather side becomes the bound; at 8x1 the matrix pipe is ~66 % busy and, MFMA and VALU
//  cycles being additive on a SIMD, the kernel sits at ~98 % of its issue bound -- DESIGN.md section 4.1)
// Ablation builds (timing only): -DCN_ABLATE_GATHER=1 2.12 ms, -DCN_ABLATE_MLP=1 1.81 ms, both 1.36 ms per C2 batch;
// -DCN_ABLATE_GATHER_LEVELS=4 / 8 (the coarsest 4 / 8 levels read no table entry and hash nothing): 2.573 / 2.490 ms against
// 2.597 -- the bound on what a cheaper (e.g. cell-major, one 64-byte line per cell) copy of the coarse levels could give
// against 2.56 ms for the real thing.
// rays per team of the team gather (fp16 mode, half table): 2 = two neighbouring pixels of a row in a lane pair; 4 = a 2 x 2
// pixel block in a lane quad, each gather wave taking 8 of the half-step's 32 samples for all four (default).  Measured at C2
// (ms per batch, mean over the ten batches): 2 rays 1.60, 4 consecutive pixels of a row 1.48, 2 x 2 block 1.39; the 2 x 2
// walk alone with 2-ray teams 1.55, and on the exact-fp32 kernel (no team gather) it costs 1.5 %, so only the team kernel walks
// its stripes that way
#ifndef CN_TEAM_RAYS
#define CN_TEAM_RAYS 4
#endif
#ifndef CN_TEAM_RAYS_PS  // the same for per-sample outputs; 512 x 3 000-sample export calls: no team 8.8, 2-ray teams 8.3,
#define CN_TEAM_RAYS_PS 1  // 4-ray teams 7.4 Gsamples/s -- consecutive samples of the exporters' rays are as close as their rays
#endif
// The team gather serves the fp16 and the split-bf16 matrix modes on either table type -- C2, ms per batch without / with
// 4-ray teams: fp16 mode on the fp32 torch table 2.14 / 1.69, split-bf16 on it 1.81 / 1.71, split-bf16 on a tcnn fp16 table
// 2.06 / 1.57 -- but NOT the exact-fp32 kernel, which is bound by SIMD issue, hides its gathers under the fp32 MFMAs and only
// pays for the per-lane ray parameters: 2.59 / 2.78 (CN_TEAM_ALL=1 forces it there too, A/B).
#ifndef CN_TEAM_XPAIR
#define CN_TEAM_XPAIR 1
#endif
#ifndef CN_TEAM_ALL
#define CN_TEAM_ALL 0
#endif
#ifndef CN_TEAM_ROW_WALK  // 1: the four rays of a team are four consecutive pixels of a row (A/B)
#define CN_TEAM_ROW_WALK 0
#endif
#ifndef CN_SPLIT_G
#define CN_SPLIT_G 8
#endif
#ifndef CN_SPLIT_MPG
#define CN_SPLIT_MPG 1
#endif
constexpr int SPLIT_G = CN_SPLIT_G, SPLIT_MPG = CN_SPLIT_MPG;
constexpr int SPLIT_PAIRS = SPLIT_G * SPLIT_MPG;
constexpr int SPLIT_THREADS = (SPLIT_G + SPLIT_PAIRS) * 64;
static_assert(SPLIT_THREADS <= 1024, "at most 16 waves per workgroup");
// one half-step in the ring: per lane 16 feature floats + selector bits
constexpr int XCH_FLOATS = 16 * 64 + 64;
constexpr int PAIR_SCRATCH = 2 * XCH_FLOATS + 64 + 68 + 68 + 4;  // ring | per-ray colour bias | matrix edges | gather edges | flags
constexpr int PAIR_FLAGS = 2 * XCH_FLOATS + 64 + 68 + 68;
constexpr size_t SPLIT_LDS_BYTES = (size_t)(BLOB_FLOATS + SPLIT_PAIRS * PAIR_SCRATCH) * sizeof(float);
constexpr size_t SPLIT_LDS_BYTES_BF16 = SPLIT_LDS_BYTES + (size_t)BF16_EXT_FLOATS * sizeof(float);
// PACK (two 48-sample rays in three half-steps, below): per pair the second ray's colour bias, matrix-side and gather-side
// bin edges, behind the pair scratch
constexpr int PACK_SCRATCH = 64 + 68 + 68;
constexpr size_t SPLIT_LDS_BYTES_PACK = SPLIT_LDS_BYTES + (size_t)(SPLIT_PAIRS * PACK_SCRATCH) * sizeof(float);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 values -> bf16 hi + bf16 lo (x ~ hi + lo to ~16 mantissa bits): the B operands of the split-bf16 matrix path
__device__ __forceinline__ void split_bf16(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? a[j] : b[j - 4];
    const __bf16 h = (__bf16)x;
    hi[j] = h;
    lo[j] = (__bf16)(x - (float)h);
  }
}
// acc += A . B with A = ah + al, B = bh + bl, dropping al . bl (v_mfma_f32_16x16x32_bf16, fp32 accumulation)
__device__ __forceinline__ f32x4 mfma_split(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  return acc;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
// eight fp32 values -> the fp16 B operand of v_mfma_f32_16x16x32_f16 (round to nearest even, as __float2half_rn)
__device__ __forceinline__ f16x8 cvt_f16x8(const f32x4& a, const f32x4& b) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[j] = (_Float16)a[j];
    r[4 + j] = (_Float16)b[j];
  }
  return r;
}
__device__ __forceinline__ f32x4 mfma_f16(const f16x8& a, const f16x8& b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}

struct SplitRay {
  float ox, oy, oz, dx, dy, dz, sn, sf;
  const float* bins;
  long long r;
  int chunk;  // per-sample outputs: the 64-sample chunk this work item covers
  bool valid;
};

// ray of schedule slot q for this pair (same mapping as render_fused_kernel).  QUAD: a column stripe is walked in 2 x 2
// pixel blocks -- four consecutive slots are one block, (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1) -- instead of pixel by
// pixel along its rows: the four-ray team gather puts a block into a lane quad.
template <bool QUAD = false>
__device__ __forceinline__ void split_ray_setup(const FusedArgs& A, long long q, long long items, int xcd, bool striped,
                                                long long rows, int cw, long long first_row, long long per_xcd,
                                                int chunks_per_ray, SplitRay& ray) {
  ray.valid = false;
  ray.chunk = 0;
  if (q >= items) return;
  long long rr;
  if (striped) {
    if constexpr (QUAD) {
    const int cwb = (cw + 1) >> 1;
    const long long per_stripe = ((rows + 1) >> 1) * cwb * 4;
    const long long sq = q / per_stripe;
    const long long qq = q - sq * per_stripe;
    const long long b = qq >> 2;
    const int w = (int)(qq & 3);
    const long long vrow = 2 * (b / cwb) + (w >> 1);
    const int cx = 2 * (int)(b % cwb) + (w & 1);
    const int col = (int)(sq * 8 + xcd) * cw + cx;
    rr = (first_row + vrow) * A.image_width + col - A.pixel_start;
    if (cx >= cw || vrow >= rows || col >= A.image_width || rr < 0 || rr >= A.num_rays) return;
    } else {
    const long long sq = q / (rows * cw);
    const long long qq = q - sq * rows * cw;
    const long long vrow = qq / cw;
    const int col = (int)(sq * 8 + xcd) * cw + (int)(qq - vrow * cw);
    rr = (first_row + vrow) * A.image_width + col - A.pixel_start;
    if (col >= A.image_width || rr < 0 || rr >= A.num_rays) return;
    }
  } else {
    rr = xcd * per_xcd + q;
  }
  if (chunks_per_ray > 0) {  // work items are (ray, chunk) pairs
#ifdef CN_PER_SAMPLE_RAY_MAJOR  // round-1 order: consecutive items = consecutive chunks of one ray
    ray.chunk = (int)(rr % chunks_per_ray);
    rr /= chunks_per_ray;
#else
    // chunk-major: consecutive items -- the pairs of a workgroup, the adjacent lanes of a team gather -- are the SAME chunk
    // of consecutive rays (neighbouring points of the exporters' surface grid: they share grid cells), not chunks 64
    // samples apart on one ray
    ray.chunk = (int)(rr / A.num_rays);
    rr -= (long long)ray.chunk * A.num_rays;
#endif
  }
  const long long r = __builtin_amdgcn_readfirstlane((int)rr);
  ray.r = r;
  ray.ox = A.origins[3 * r];
  ray.oy = A.origins[3 * r + 1];
  ray.oz = A.origins[3 * r + 2];
  ray.dx = A.directions[3 * r];
  ray.dy = A.directions[3 * r + 1];
  ray.dz = A.directions[3 * r + 2];
  ray.sn = spacing_fn(A.spacing, A.nears[r]);
  ray.sf = spacing_fn(A.spacing, A.fars[r]);
  ray.bins = A.bins ? A.bins + r * (long long)(A.S + 1) : nullptr;
  ray.valid = true;
}

__device__ __forceinline__ float split_edge(const FusedArgs& A, const SplitRay& ray, int i) {
  i = min(i, A.S);
  return ray.bins ? ray.bins[i] : spacing_to_euclid(A.spacing, linspace01(i, A.S + 1), ray.sn, ray.sf);
}

__device__ __forceinline__ void split_fill_edges(const FusedArgs& A, const SplitRay& ray, int c0, float* tb, int lane) {
  const float e_lo = split_edge(A, ray, c0 + lane);
  const float e_top = split_edge(A, ray, c0 + 64);
  __builtin_amdgcn_wave_barrier();
  tb[lane] = e_lo;
  if (lane == 0) tb[64] = e_top;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// MM (cn_render_opts.matrix_precision): MM_BF16 -- the matrix waves run the MLPs on split-bf16 products (the weight images in
// the blob are then the bf16 ones of prep_kernel, four more blocks of them behind the pair scratch); MM_F16 -- on fp16
// products (v_mfma_f32_16x16x32_f16, fp32 accumulation; tcnn's FullyFusedMLP arithmetic): the gather waves hand over fp16
// features (16 bytes per lane and sample instead of 32), blended on packed fp16 pairs when the table is a half table.
// HALF: half2 table entries (CN_TABLE_F16).  GENERIC: per-level index records (tcnn layout), see lane_level_rec.
// PACK (exact-fp32 products, 32 < S <= 48 -- the default method's 48 field samples per ray, nerfacto's
// num_nerf_samples_per_ray inherited at fruit_nerf.py:59-68): a ray is THREE 16-sample column tiles and a half-step is two, so a
// ray per two half-steps leaves every fourth tile empty -- a quarter of the matrix waves' MFMAs and of the gather waves' table
// reads spent on samples past the end of the ray.  Here a pair walks its schedule slots two rays at a time, A and B, in three
// half-steps: (A0, A1), (A2, B0), (B1, B2).  The two-tile MLP chain is the same code (a copy of it for one tile next to it
// makes hipcc spill ~50 of the 128 registers a 1024-thread workgroup leaves per lane: built three ways, 0.96 vs 0.91 ms per
// 65 536 rays, removed); what changes is which ray a tile belongs to -- its sample positions on the gather side, its per-ray
// colour bias, its lanes in the 64-lane compositing row (tile B0 is computed in A's last half-step and moved from lanes 48..63
// to lanes 0..15 when B's row starts) on the matrix side.  Per ray the arithmetic, hence every output bit, is unchanged.
template <bool PER_SAMPLE, int MM = MM_FP32, bool HALF = false, bool GENERIC = false, bool PACK = false>
__global__ void __launch_bounds__(SPLIT_THREADS) render_split_kernel(FusedArgs A) {
  static_assert(!PACK || MM == MM_FP32, "the packed schedule exists for the exact-fp32 kernel (the other modes gather in teams)");
  constexpr bool BF16 = MM == MM_BF16, F16 = MM == MM_F16;
  // TEAM (fp16 or split-bf16 products): the gather waves of neighbouring pairs work as a team -- see the gather role
  // rays (= pairs = gather waves) per team: 2 or 4 (1 = no team).  Per-sample outputs (the exporters' parallel rays, whose
  // neighbouring SAMPLES are as close as neighbouring rays) keep 16 consecutive samples of one ray per lane row: no team
  constexpr int TR = PER_SAMPLE ? CN_TEAM_RAYS_PS : CN_TEAM_RAYS;
  constexpr bool TEAM = !PACK && (CN_TEAM_ALL || MM != MM_FP32) && TR > 1 && (SPLIT_MPG == 1) && (SPLIT_G % TR == 0) && !CN_ABLATE_GATHER;
  constexpr bool QUAD = TEAM && TR == 4 && !CN_TEAM_ROW_WALK;  // stripes walked in 2 x 2 pixel blocks (split_ray_setup)
  extern __shared__ __align__(16) float lds[];
  constexpr int OFF_EXT = BLOB_FLOATS + SPLIT_PAIRS * PAIR_SCRATCH;  // BF16 only
  {
    const float4* src = reinterpret_cast<const float4*>(A.blob);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < BLOB_FLOATS / 4; i += SPLIT_THREADS) dst[i] = src[i];
    if constexpr (BF16) {
      const float4* esrc = reinterpret_cast<const float4*>(A.blob_ext);
      float4* edst = reinterpret_cast<float4*>(lds + OFF_EXT);
      for (int i = threadIdx.x; i < BF16_EXT_FLOATS / 4; i += SPLIT_THREADS) edst[i] = esrc[i];
    }
    if (threadIdx.x < SPLIT_PAIRS * 4)
      reinterpret_cast<int*>(lds + BLOB_FLOATS + (threadIdx.x >> 2) * PAIR_SCRATCH + PAIR_FLAGS)[threadIdx.x & 3] =
          (threadIdx.x & 3) == 2 ? -1 : 0;  // [2]: schedule slot of a ray its matrix wave has terminated early
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane0 = lane_id();
  const bool matrix_role = wave >= SPLIT_G;
  const int pair = matrix_role ? wave - SPLIT_G : wave * SPLIT_MPG;  // matrix wave: its own; gather wave: its first
  float* ps = lds + BLOB_FLOATS + pair * PAIR_SCRATCH;
  float* ring = ps;
  float* scratch = ps + 2 * XCH_FLOATS;  // per-ray colour bias (written and read by the matrix wave only)
  float* tb_m = scratch + 64;
  // PACK: ray B's colour bias | matrix-side edges | gather-side edges
  float* packs = lds + BLOB_FLOATS + SPLIT_PAIRS * PAIR_SCRATCH + pair * PACK_SCRATCH;
  const int S = A.S;

  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const long long stride = (long long)(gridDim.x >> 3) * SPLIT_PAIRS;
  // Per-sample outputs have no dependency between the chunks of a ray, so the work items are (ray, 64-sample chunk)
  // pairs: the exporters' 512-ray x 3000-sample calls then fill the device (24 064 items instead of 512).
  const int nchunks = (S + 63) >> 6;
  const int chunks_per_item_ray = PER_SAMPLE ? nchunks : 0;
  const long long nwork = PER_SAMPLE ? A.num_rays * nchunks : A.num_rays;
  const bool striped = !PER_SAMPLE && A.image_width > 0;
  const long long per_xcd = (nwork + 7) >> 3;
  const int nstripe = 8 * A.stripes_per_xcd;
  const int cw = striped ? (A.image_width + nstripe - 1) / nstripe : 0;
  const long long first_row = striped ? A.pixel_start / A.image_width : 0;
  const long long last_row = striped ? (A.pixel_start + A.num_rays - 1) / A.image_width : 0;
  const long long rows = last_row - first_row + 1;
  const long long items =
      !striped ? min(per_xcd, max(nwork - xcd * per_xcd, 0LL))
               : (QUAD ? ((rows + 1) >> 1) * ((cw + 1) >> 1) * 4 : rows * cw) * A.stripes_per_xcd;

  // every wave of the workgroup runs the same number of half-steps (the barrier count must match): the schedule slots
  // of pair 0, the longest list; a slot without a ray (past the end, outside the image) is an idle step
  const long long q_first = (long long)slot * SPLIT_PAIRS;
  const long long n_q = q_first < items ? (items - q_first + stride - 1) / stride : 0;
  const int nhalf = PACK ? 3 : (PER_SAMPLE ? 2 : 2 * nchunks);
  const long long total = PACK ? ((n_q + 1) >> 1) * 3 : n_q * nhalf;  // PACK: three half-steps per two schedule slots

  SplitRay ray;
  ray.valid = false;
  SplitRay ray_b;  // PACK: the second ray of the duo (matrix wave) ...
  ray_b.valid = false;
  SplitRay gray_b;  // ... and of the gather wave
  gray_b.valid = false;
  SplitRay gray[SPLIT_MPG];  // gather wave: the rays of the matrix waves it feeds

#pragma unroll
  for (int m = 0; m < SPLIT_MPG; ++m) gray[m].valid = false;
  CompositeState st;
  float my_dlogit = 0.f, my_sel = 0.f, my_sem = 0.f, my_r = 0.f, my_g = 0.f, my_b = 0.f;
  bool ray_stopped = false;  // matrix wave: this ray was terminated early (cn_render_opts.early_stop_transmittance)

#ifdef CN_SPLIT_GATHER_PRIO  // A/B: gather waves win the VALU arbitration, so their loads are issued before the matrix waves' MLPs run
  if (!matrix_role) __builtin_amdgcn_s_setprio(CN_SPLIT_GATHER_PRIO);
#endif
  // One loop for both roles (same trip count, one workgroup barrier per half-step at its end).
  for (long long step = 0; step <= total; ++step) {
    if (!matrix_role) {
      // ================= gather wave: produce half-step `step` =======================================================
      // (lane coordinates through an empty volatile asm per iteration: keeps LICM from hoisting every lane-dependent
      //  LDS address out of the loop and spilling them)
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int g = lane >> 4, j = lane & 15;
      if constexpr (TEAM) {
        // ---- team gather -------------------------------------------------------------------------------------------------
        // With fp16 products the kernel runs at the rate the CU's L1 looks up cache lines for per-lane-addressed loads: one
        // lane per clock, TWO when the lanes of a pair (2k, 2k + 1) of an 8-byte load fall into one line, four when a whole
        // quad does (tools/gather_rate_microbench.hip).  Consecutive samples of ONE ray are 5e-3 of the box apart and share
        // cells on the coarsest levels only; the same sample of two NEIGHBOURING pixels is 5e-4 apart and shares its cell --
        // hence its table entries -- up to level 12 or so.  So the gather waves of pairs 2T and 2T + 1 (neighbouring pixels:
        // consecutive schedule slots) split the work the other way round: each handles 16 of the half-step's 32 samples
        // (one column tile) for BOTH rays, the two rays in adjacent lanes, and writes the features into the owning pair's
        // ring in the layout its matrix wave reads.  Arithmetic per (ray, sample) is unchanged.
        if (step < total) {
          const long long qi = step / nhalf;
          const int k = (int)(step - qi * nhalf);
          constexpr int SPW = 32 / TR;  // samples of the half-step one gather wave takes (for all TR rays): 16 or 8
          const int pair0 = wave & ~(TR - 1), sub = wave & (TR - 1);
          // the rays' parameters live in this wave's own (otherwise unused) gather-edge area of LDS, 16 dwords per ray
          // {o, d, s(near), s(far), bins pointer, chunk, valid}: kept in scalar registers for a whole ray they spill
          float* trec = ring + 2 * XCH_FLOATS + 64 + 68;
          static_assert(TR * 16 <= 68, "ray records fit the gather-edge area");
          if (k == 0) {
#pragma unroll
            for (int t = 0; t < TR; ++t) {
              SplitRay tr;
              split_ray_setup<QUAD>(A, q_first + pair0 + t + qi * stride, items, xcd, striped, rows, cw, first_row, per_xcd,
                              chunks_per_item_ray, tr);
              __builtin_amdgcn_wave_barrier();
              if (lane == 0) {
                float* w = trec + 16 * t;
                const unsigned long long bp = reinterpret_cast<unsigned long long>(tr.bins);
                w[0] = tr.ox; w[1] = tr.oy; w[2] = tr.oz; w[3] = tr.dx;
                w[4] = tr.dy; w[5] = tr.dz; w[6] = tr.sn; w[7] = tr.sf;
                reinterpret_cast<unsigned*>(w)[8] = (unsigned)bp;
                reinterpret_cast<unsigned*>(w)[9] = (unsigned)(bp >> 32);
                reinterpret_cast<int*>(w)[10] = tr.chunk;
                reinterpret_cast<int*>(w)[11] = tr.valid ? 1 : 0;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
          float* ring0 = lds + BLOB_FLOATS + pair0 * PAIR_SCRATCH;
          const int mcol = lane & 15, rsel = mcol & (TR - 1), sidx = mcol / TR;
          bool any = false, active = false;
#pragma unroll
          for (int t = 0; t < TR; ++t) {
            const bool a = reinterpret_cast<const int*>(trec + 16 * t)[11] != 0 &&
                           !(!PER_SAMPLE && A.early_stop > 0.f &&
                             reinterpret_cast<volatile int*>(ring0 + t * PAIR_SCRATCH + PAIR_FLAGS)[2] == (int)qi);
            any = any || a;
            active = rsel == t ? a : active;
          }
          // (Skipping the team's waves whose columns lie past the end of a ray -- S = 48: the last 16 columns of every second
          //  half-step; zeros handed over, nothing read -- and the same skip in the one-ray branch below were both built and
          //  measured on one box: the 48-sample eval image does not move (11.30 vs 11.29 ms in fp16 mode, 15.28 vs 15.29 in fp32)
          //  and the 192-sample batch pays for the extra path, 1.417 vs 1.385 ms in fp16 mode, 2.68 vs 2.59 ms in fp32.  Not built in.)
          if (any) {
            if (active) {
              const f32x4 ra = *reinterpret_cast<const f32x4*>(trec + 16 * rsel);
              const f32x4 rb = *reinterpret_cast<const f32x4*>(trec + 16 * rsel + 4);
              const u32x4v rc = *reinterpret_cast<const u32x4v*>(trec + 16 * rsel + 8);
              const float rox = ra.x, roy = ra.y, roz = ra.z, rdx = ra.w, rdy = rb.x, rdz = rb.y, rsn = rb.z, rsf = rb.w;
              const float* rbins = reinterpret_cast<const float*>((unsigned long long)rc.x | ((unsigned long long)rc.y << 32));
              const int chunk = PER_SAMPLE ? (int)rc.z : (k >> 1), half = k & 1;
              auto edge = [&](int i) -> float {
                i = min(i, S);
                return rbins ? rbins[i] : spacing_to_euclid(A.spacing, linspace01(i, S + 1), rsn, rsf);
              };
              float px[2], py[2], pz[2];
              bool sel[2];
              int col[2];  // column of the half-step (0..31) of this lane's two samples
#pragma unroll
              for (int h = 0; h < 2; ++h) {  // this wave's samples: SPW sub + (SPW / 2) h + sidx of the half-step
                col[h] = SPW * sub + (SPW / 2) * h + sidx;
                const int i0 = chunk * 64 + 32 * half + col[h];
                const float mid = (edge(i0) + edge(i0 + 1)) / 2.f;
                px[h] = rox + rdx * mid;
                py[h] = roy + rdy * mid;
                pz[h] = roz + rdz * mid;
                sel[h] = normalize_position(A.scene, px[h], py[h], pz[h]);
              }
              u32x4v featp[2];
              f32x4 feat[2][2];
              const f32x4 lvl_scale = *reinterpret_cast<const f32x4*>(lds + OFF_SCALE + 4 * g);
              const float pos_off = GENERIC ? A.grid.pos_offset : 0.f;
              if constexpr (F16 && HALF) {
                PkLoads cur = hash_level_pk_issue<GENERIC>(A.grid.table, lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g, lvl_scale[0]),
                                                           pos_off, px[0], py[0], pz[0]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  PkLoads nxt;
                  if (u < 7) {
                    const int qn = (u + 1) >> 1, hn = (u + 1) & 1;
                    nxt = hash_level_pk_issue<GENERIC>(A.grid.table,
                                                       lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g + qn, lvl_scale[qn]), pos_off,
                                                       px[hn], py[hn], pz[hn]);
                  }
                  featp[u & 1][u >> 1] = pk_pin(hash_level_pk_blend(cur));
                  if (u < 7) cur = nxt;
                }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const Lvl lv = lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g + q, lvl_scale[q]);
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
#if CN_TEAM_XPAIR  // aligned 16-byte (float table) / 8-byte (half table) pair gathers for the fp32 blend too: split-bf16 on the
                    // headline table 1.726 -> 1.686 ms (the fp16 blend of a half table has its own pair form above)
                    const float2 f = hash_level_xpair<HALF>(A.grid.table, lv, pos_off, px[h], py[h], pz[h]);
#else
                    const float2 f = hash_level_sc<HALF, GENERIC>(A.grid.table, lv, pos_off, px[h], py[h], pz[h]);
#endif
                    if constexpr (F16) {
                      const f16x2 hp = {(_Float16)f.x, (_Float16)f.y};
                      featp[h][q] = __builtin_bit_cast(unsigned, hp);
                    } else {
                      if (q == 0) { feat[h][0].x = f.x; feat[h][0].y = f.y; }
                      if (q == 1) { feat[h][0].z = f.x; feat[h][0].w = f.y; }
                      if (q == 2) { feat[h][1].x = f.x; feat[h][1].y = f.y; }
                      if (q == 3) { feat[h][1].z = f.x; feat[h][1].w = f.y; }
                    }
                  }
                  if ((q + 1) % CN_SPLIT_LEVELS_IN_FLIGHT == 0) __builtin_amdgcn_sched_barrier(0);
                }
              }
              float* xs = ring0 + rsel * PAIR_SCRATCH + (int)(step & 1) * XCH_FLOATS;
#pragma unroll
              for (int h = 0; h < 2; ++h) {  // column c of the half-step = column tile c >> 4, column c & 15 of the matrix wave
                if constexpr (F16) {
                  reinterpret_cast<f16x8*>(xs)[(col[h] >> 4) * 64 + 16 * g + (col[h] & 15)] = __builtin_bit_cast(f16x8, featp[h]);
                } else {
                  f32x4* xv = reinterpret_cast<f32x4*>(xs);
                  xv[(2 * (col[h] >> 4) + 0) * 64 + 16 * g + (col[h] & 15)] = feat[h][0];
                  xv[(2 * (col[h] >> 4) + 1) * 64 + 16 * g + (col[h] & 15)] = feat[h][1];
                }
                if (g == 0) xs[XCH_FLOATS - 64 + col[h]] = sel[h] ? 1.f : 0.f;  // selector of that column
              }
            }
          }
        }
      } else
      if (step < total) {
        const long long qi = step / nhalf;
        const int k = (int)(step - qi * nhalf);
#pragma unroll
        for (int m = 0; m < SPLIT_MPG; ++m) {
          SplitRay& gr = gray[m];
          float* gring = ring + m * PAIR_SCRATCH;
          float* tb_g = gring + 2 * XCH_FLOATS + 64 + 68;
          if constexpr (PACK) {
            // duo qi = schedule slots 2 qi (ray A) and 2 qi + 1 (ray B): A is set up in the duo's first half-step, B in its second
            // (under the striped schedule a slot in the middle of a pair's list can be empty: either ray may be missing)
            if (k == 0) {
              split_ray_setup<QUAD>(A, q_first + pair + m + 2 * qi * stride, items, xcd, striped, rows, cw, first_row, per_xcd, chunks_per_item_ray, gr);
              if (gr.valid) split_fill_edges(A, gr, (PER_SAMPLE ? gr.chunk : 0) * 64, tb_g, lane);
              gray_b.valid = false;
            }
            if (k == 1) {
              split_ray_setup<QUAD>(A, q_first + pair + m + (2 * qi + 1) * stride, items, xcd, striped, rows, cw, first_row, per_xcd, chunks_per_item_ray, gray_b);
              if (gray_b.valid) split_fill_edges(A, gray_b, (PER_SAMPLE ? gray_b.chunk : 0) * 64, packs + 64 + 68, lane);
            }
          } else {
          if (k == 0)
            split_ray_setup<QUAD>(A, q_first + pair + m + qi * stride, items, xcd, striped, rows, cw, first_row, per_xcd, chunks_per_item_ray, gr);
          }
          // early termination: the matrix wave publishes the schedule slot of a ray it has finished early; the rest of
          // that ray's half-steps are then idle for the pair (the workgroup still runs them in lock-step: the time is
          // saved when the 8 rays in flight -- neighbouring pixels -- go opaque at about the same depth)
          const bool stopped_g = !PACK && !PER_SAMPLE && A.early_stop > 0.f &&
                                 reinterpret_cast<volatile int*>(gring + PAIR_FLAGS)[2] == (int)qi;
          // PACK: the half-step is live when a ray that owns one of its tiles exists
          const bool live_g = PACK ? (k == 0 ? gr.valid : (k == 1 ? (gr.valid || gray_b.valid) : gray_b.valid)) : gr.valid;
          if (live_g && !stopped_g) {
            const int chunk = PER_SAMPLE ? gr.chunk : (k >> 1), half = k & 1;
            if constexpr (!PACK) {
              if (half == 0) split_fill_edges(A, gr, chunk * 64, tb_g, lane);
            }
            float px[2], py[2], pz[2];
            bool sel[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if constexpr (PACK) {
                // tile c of half-step k: (A0, A1), (A2, B0), (B1, B2); the tile of a missing ray is computed on the other (unused)
                const bool owner_b = c == 0 ? k == 2 : k >= 1;
                const bool of_b = owner_b ? gray_b.valid : !gr.valid;
                const int t16 = c == 0 ? (k == 0 ? 0 : (k == 1 ? 32 : 16)) : (k == 0 ? 16 : (k == 1 ? 0 : 32));
                const float* tb = of_b ? packs + 64 + 68 : tb_g;
                const int kk = t16 + j;
                const float mid = (tb[kk] + tb[kk + 1]) / 2.f;
                px[c] = (of_b ? gray_b.ox : gr.ox) + (of_b ? gray_b.dx : gr.dx) * mid;
                py[c] = (of_b ? gray_b.oy : gr.oy) + (of_b ? gray_b.dy : gr.dy) * mid;
                pz[c] = (of_b ? gray_b.oz : gr.oz) + (of_b ? gray_b.dz : gr.dz) * mid;
              } else {
              const int kk = 32 * half + 16 * c + j;
              const float mid = (tb_g[kk] + tb_g[kk + 1]) / 2.f;
              px[c] = gr.ox + gr.dx * mid;
              py[c] = gr.oy + gr.dy * mid;
              pz[c] = gr.oz + gr.dz * mid;
              }
              sel[c] = normalize_position(A.scene, px[c], py[c], pz[c]);
            }
            f32x4 feat[2][2];
            u32x4v featp[2];  // F16: the packed fp16 pairs of the four levels = the MFMA B operand as it stands
            const f32x4 lvl_scale = *reinterpret_cast<const f32x4*>(lds + OFF_SCALE + 4 * g);
            if constexpr (F16 && HALF && !CN_ABLATE_GATHER) {
              // pair gathers (hash_level_pk_issue): the eight (level, sample) units in program order, unit n + 1's loads
              // issued before unit n is blended
              const float pos_off = GENERIC ? A.grid.pos_offset : 0.f;
              PkLoads cur = hash_level_pk_issue<GENERIC>(A.grid.table, lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g, lvl_scale[0]),
                                                         pos_off, px[0], py[0], pz[0]);
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int q = u >> 1, c = u & 1;
                PkLoads nxt;
                if (u < 7) {
                  const int qn = (u + 1) >> 1, cn_ = (u + 1) & 1;
                  nxt = hash_level_pk_issue<GENERIC>(A.grid.table,
                                                     lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g + qn, lvl_scale[qn]), pos_off,
                                                     px[cn_], py[cn_], pz[cn_]);
                }
                featp[c][q] = pk_pin(hash_level_pk_blend(cur));
                if (u < 7) cur = nxt;
              }
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const Lvl lv = lane_level_rec<GENERIC>(lds + OFF_LVL, A.grid, 4 * g + q, lvl_scale[q]);
              const float pos_off = GENERIC ? A.grid.pos_offset : 0.f;
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                if constexpr (F16 && HALF) {  // (timing-only ablation build)
                  featp[c][q] = __builtin_bit_cast(unsigned, px[c] * lv.scale + py[c]) ^ __builtin_bit_cast(unsigned, pz[c]);
                  continue;
                }
#if CN_ABLATE_GATHER  // timing-only build: no table reads
                const float2 f = make_float2(px[c] * lv.scale, py[c] + pz[c]);
#elif defined(CN_ABLATE_GATHER_LEVELS)  // timing-only build: levels below CN_ABLATE_GATHER_LEVELS read no table entry
                const float2 f = (4 * g + q < CN_ABLATE_GATHER_LEVELS)
                                     ? make_float2(px[c] * lv.scale, py[c] + pz[c])
                                     : hash_level_sc<HALF, GENERIC>(A.grid.table, lv, pos_off, px[c], py[c], pz[c]);
#else
                const float2 f = hash_level_sc<HALF, GENERIC>(A.grid.table, lv, pos_off, px[c], py[c], pz[c]);
#endif
                if constexpr (F16) {  // float table: blended in fp32, handed over as fp16
                  f16x2 hp = {(_Float16)f.x, (_Float16)f.y};
                  featp[c][q] = __builtin_bit_cast(unsigned, hp);
                  continue;
                }
                if (q == 0) { feat[c][0].x = f.x; feat[c][0].y = f.y; }
                if (q == 1) { feat[c][0].z = f.x; feat[c][0].w = f.y; }
                if (q == 2) { feat[c][1].x = f.x; feat[c][1].y = f.y; }
                if (q == 3) { feat[c][1].z = f.x; feat[c][1].w = f.y; }
              }
              if ((q + 1) % CN_SPLIT_LEVELS_IN_FLIGHT == 0)
                __builtin_amdgcn_sched_barrier(0);  // bound the gathers in flight (16 per lane and level)
            }
            }
            float* xs = gring + (int)(step & 1) * XCH_FLOATS;
            if constexpr (F16) {
              f16x8* xh = reinterpret_cast<f16x8*>(xs);
#pragma unroll
              for (int c = 0; c < 2; ++c)
                xh[c * 64 + lane] = __builtin_bit_cast(f16x8, featp[c]);
            } else {
              f32x4* xv = reinterpret_cast<f32x4*>(xs);
              xv[0 * 64 + lane] = feat[0][0];
              xv[1 * 64 + lane] = feat[0][1];
              xv[2 * 64 + lane] = feat[1][0];
              xv[3 * 64 + lane] = feat[1][1];
            }
            xs[XCH_FLOATS - 64 + lane] = (sel[0] ? 1.f : 0.f) + (sel[1] ? 2.f : 0.f);
          }
          __builtin_amdgcn_sched_barrier(0);  // one consumer's half-step at a time
        }
      }
    } else if (step >= 1) {
      // ================= matrix wave: consume half-step `step - 1` ===================================================
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int g = lane >> 4, j = lane & 15;
      const long long hs = step - 1;
      const long long qi = hs / nhalf;
      const int k = (int)(hs - qi * nhalf);
      if (PACK ? k <= 1 : k == 0) {
        // PACK: k = 0 sets up ray A of the duo (schedule slot 2 qi), k = 1 ray B (slot 2 qi + 1) -- one copy of the code,
        // the ray record and the LDS areas picked by k
        SplitRay cur;
        split_ray_setup<QUAD>(A, q_first + pair + (PACK ? 2 * qi + k : qi) * stride, items, xcd, striped, rows, cw, first_row, per_xcd, chunks_per_item_ray, cur);
        if (!PACK || k == 0) {
          ray = cur;
          st = CompositeState();
          ray_stopped = false;
          ray_b.valid = false;
        } else {
          ray_b = cur;
        }
        float* scratch_k = (PACK && k == 1) ? packs : scratch;
        if constexpr (PACK) {
          if (cur.valid) split_fill_edges(A, cur, (PER_SAMPLE ? cur.chunk : 0) * 64, k == 1 ? packs + 64 : tb_m, lane);
          if (k == 0) my_dlogit = my_sel = my_sem = my_r = my_g = my_b = 0.f;
        }
        if (cur.valid) {
          // per-ray colour bias: bc0 + Wc0[:, sh].SH(d) + Wc0[:, app].app  (lane n = neuron n)
          float sx = cur.dx, sy = cur.dy, sz = cur.dz;
          if (!A.sh_unit) {
            sx = (sx + 1.f) / 2.f;
            sy = (sy + 1.f) / 2.f;
            sz = (sz + 1.f) / 2.f;
          }
          float sh[16];
          sh_deg4(sx, sy, sz, sh);
          if constexpr (F16) {  // the colour network's inputs are fp16 values (tcnn casts them)
#pragma unroll
            for (int q = 0; q < 16; ++q) sh[q] = (float)(_Float16)sh[q];
          }
          const long long row = A.app_per_camera ? A.cam_idx[cur.r] : 0;
          float bias = A.app_bias[row * 64 + lane];
          const f32x4* wsh = reinterpret_cast<const f32x4*>(lds + OFF_WSH + lane * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w = wsh[q];
            bias = fmaf(w.x, sh[4 * q + 0], bias);
            bias = fmaf(w.y, sh[4 * q + 1], bias);
            bias = fmaf(w.z, sh[4 * q + 2], bias);
            bias = fmaf(w.w, sh[4 * q + 3], bias);
          }
          __builtin_amdgcn_wave_barrier();
          scratch_k[lane] = bias;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      const bool live_m = PACK ? (k == 0 ? ray.valid : (k == 1 ? (ray.valid || ray_b.valid) : ray_b.valid)) : ray.valid;
      if (live_m && !ray_stopped) {
        const int chunk = PACK ? 0 : (PER_SAMPLE ? ray.chunk : (k >> 1)), half = PACK ? (k >= 1 ? 1 : 0) : (k & 1);
        const int c0 = chunk * 64;
        if constexpr (!PACK) {
          if (half == 0) {
            split_fill_edges(A, ray, c0, tb_m, lane);
            my_dlogit = my_sel = my_sem = my_r = my_g = my_b = 0.f;
          }
        }
        // lane group of tile c: gb + c.  Unpacked: tiles (0, 1) of half-step `half` are lane rows 2 half, 2 half + 1.  PACK:
        // (A0, A1) -> rows 0, 1; (A2, B0) -> rows 2, 3 (B0 moves to row 0 when A's row is done); (B1, B2) -> rows 1, 2
        const int gb = PACK ? (k == 0 ? 0 : (k == 1 ? 2 : 1)) : 2 * half;
        // per-ray colour bias of tile c (PACK: tile 1 of the middle half-step and both tiles of the last belong to ray B)
        const float* cbias0 = (PACK && k == 2) ? packs : scratch;
        const float* cbias1 = (PACK && k >= 1) ? packs : scratch;
        const float* xs = ring + (int)(hs & 1) * XCH_FLOATS;
        const f32x4* xv = reinterpret_cast<const f32x4*>(xs);
        f32x4 feat[2][2];
        f16x8 fb[2];  // F16: the features as the gather wave converted them
        if constexpr (F16) {
          fb[0] = reinterpret_cast<const f16x8*>(xs)[0 * 64 + lane];
          fb[1] = reinterpret_cast<const f16x8*>(xs)[1 * 64 + lane];
        } else {
          feat[0][0] = xv[0 * 64 + lane];
          feat[0][1] = xv[1 * 64 + lane];
          feat[1][0] = xv[2 * 64 + lane];
          feat[1][1] = xv[3 * 64 + lane];
        }
        // selector of this lane's own sample: the gather wave's bit pair per lane, or (TEAM) one float per column
        const int selbits = TEAM ? ((xs[XCH_FLOATS - 64 + 16 * (g & 1) + j] != 0.f) ? 3 : 0) : (int)xs[XCH_FLOATS - 64 + lane];
        // ---- base MLP (32 -> 64 ReLU -> 16) ---------------------------------------------------------------------------------
        f32x4 h[4][2];
        f32x4 o16[2];
        // image block b: [hi | lo][lane][8 bf16]; 18 blocks sit in the fp32 A region, 4 behind the pair scratch
        auto blk = [&](int b) { return reinterpret_cast<const bf16x8*>(b < 18 ? lds + OFF_A0 + b * 512 : lds + OFF_EXT + (b - 18) * 512); };
        auto blkh = [&](int b) { return reinterpret_cast<const f16x8*>(lds + OFF_A0 + b * 256)[lane]; };  // F16 image block
        if constexpr (F16) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_B0 + 16 * mt + 4 * g);
            const f16x8 a = blkh(mt);
#pragma unroll
            for (int c = 0; c < 2; ++c) h[mt][c] = relu4(mfma_f16(a, fb[c], b));
          }
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + 4 * g);
          f32x4 acc[2] = {b1, b1};
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const f16x8 a = blkh(4 + kb);
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = mfma_f16(a, cvt_f16x8(h[2 * kb][c], h[2 * kb + 1][c]), acc[c]);
          }
          o16[0] = acc[0];
          o16[1] = acc[1];
        } else if constexpr (BF16) {
          bf16x8 fh[2], fl[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) split_bf16(feat[c][0], feat[c][1], fh[c], fl[c]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_B0 + 16 * mt + 4 * g);
            const bf16x8 ah = blk(mt)[lane], al = blk(mt)[64 + lane];
#pragma unroll
            for (int c = 0; c < 2; ++c) h[mt][c] = relu4(mfma_split(ah, al, fh[c], fl[c], b));
          }
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + 4 * g);
          f32x4 acc[2] = {b1, b1};
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const bf16x8 ah = blk(4 + kb)[lane], al = blk(4 + kb)[64 + lane];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              bf16x8 xh, xl;
              split_bf16(h[2 * kb][c], h[2 * kb + 1][c], xh, xl);
              acc[c] = mfma_split(ah, al, xh, xl, acc[c]);
            }
          }
          o16[0] = acc[0];
          o16[1] = acc[1];
        } else {
          // ---- base MLP layer 0: 32 -> 64, ReLU ------------------------------------------------------------------------
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
          // ---- base MLP layer 1: 64 -> 16 ---------------------------------------------------------------------------------
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
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool mine = PACK ? (g == gb || g == gb + 1) : ((g >> 1) == half);
        const bool odd = PACK ? (g == gb + 1) : ((g & 1) != 0);
        {
          const float d0 = row0_broadcast(o16[0].x), d1 = row0_broadcast(o16[1].x);
          const float dsel = odd ? d1 : d0;
          const float ssel = (selbits & (odd ? 2 : 1)) ? 1.f : 0.f;
          my_dlogit = mine ? dsel : my_dlogit;
          my_sel = mine ? ssel : my_sel;
        }
        // ---- semantics ------------------------------------------------------------------------------------------------------
        float sem_part[2] = {0.f, 0.f};
        bf16x8 oh[2], ol[2];  // BF16: the 16 base outputs as a K block of 32 (upper half zero)
        f16x8 o16h[2];        // F16: the same
        if constexpr (BF16) {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 2; ++c) split_bf16(o16[c], zero4, oh[c], ol[c]);
        }
        if constexpr (F16) {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 2; ++c) o16h[c] = cvt_f16x8(o16[c], zero4);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_BS0 + 16 * mt + 4 * g);
          const f32x4 wf = *reinterpret_cast<const f32x4*>(lds + OFF_WF + 16 * mt + 4 * g);
          f32x4 acc[2] = {b, b};
          if constexpr (F16) {
            const f16x8 a = blkh(6 + mt);
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = mfma_f16(a, o16h[c], acc[c]);
          } else if constexpr (BF16) {
            const bf16x8 ah = blk(6 + mt)[lane], al = blk(6 + mt)[64 + lane];
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = mfma_split(ah, al, oh[c], ol[c], acc[c]);
          } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_AS0 + (mt * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], o16[c][e], acc[c]);
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) sem_part[c] = dot4(wf, relu4(acc[c]), sem_part[c]);
        }
        // ---- colour layer 0 ----------------------------------------------------------------------------------------------------
        f32x4 c1[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 cb = *reinterpret_cast<const f32x4*>(cbias0 + 16 * mt + 4 * g);
          const f32x4 cb_1 = PACK ? *reinterpret_cast<const f32x4*>(cbias1 + 16 * mt + 4 * g) : cb;
          f32x4 acc[2] = {cb, cb_1};
          if constexpr (F16) {
            const f16x8 a = blkh(10 + mt);
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = mfma_f16(a, o16h[c], acc[c]);
          } else if constexpr (BF16) {
            const bf16x8 ah = blk(10 + mt)[lane], al = blk(10 + mt)[64 + lane];
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = mfma_split(ah, al, oh[c], ol[c], acc[c]);
          } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_AC0 + (mt * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], o16[c][e], acc[c]);
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) c1[mt][c] = relu4(acc[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- colour layer 1 + rgb head -------------------------------------------------------------------------------------------
        float rgb_part[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        bf16x8 ch[2][2], cl[2][2];  // BF16: [K block][column tile] operands of colour layer 1
        f16x8 c1h[2][2];  // F16: the same
        if constexpr (BF16) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < 2; ++c) split_bf16(c1[2 * kb][c], c1[2 * kb + 1][c], ch[kb][c], cl[kb][c]);
        }
        if constexpr (F16) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < 2; ++c) c1h[kb][c] = cvt_f16x8(c1[2 * kb][c], c1[2 * kb + 1][c]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_BC1 + 16 * mt + 4 * g);
          f32x4 acc[2] = {b, b};
          if constexpr (F16) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              const f16x8 a = blkh(14 + 2 * mt + kb);
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = mfma_f16(a, c1h[kb][c], acc[c]);
            }
          } else if constexpr (BF16) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              const bf16x8 ah = blk(14 + 2 * mt + kb)[lane], al = blk(14 + 2 * mt + kb)[64 + lane];
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = mfma_split(ah, al, ch[kb][c], cl[kb][c], acc[c]);
            }
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(lds + OFF_AC1 + ((mt * 4 + t) * 64 + lane) * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c] = MFMA(a[e], c1[t][c][e], acc[c]);
            }
          }
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds + OFF_WRGB + 0 * 64 + 16 * mt + 4 * g);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(lds + OFF_WRGB + 1 * 64 + 16 * mt + 4 * g);
          const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds + OFF_WRGB + 2 * 64 + 16 * mt + 4 * g);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x4 v = relu4(acc[c]);
            rgb_part[c][0] = dot4(w0, v, rgb_part[c][0]);
            rgb_part[c][1] = dot4(w1, v, rgb_part[c][1]);
            rgb_part[c][2] = dot4(w2, v, rgb_part[c][2]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const float s0 = group_sum(sem_part[0]), s1 = group_sum(sem_part[1]);
          my_sem = mine ? (odd ? s1 : s0) : my_sem;
          const float r0 = group_sum(rgb_part[0][0]), r1 = group_sum(rgb_part[1][0]);
          my_r = mine ? (odd ? r1 : r0) : my_r;
          const float g0 = group_sum(rgb_part[0][1]), g1 = group_sum(rgb_part[1][1]);
          my_g = mine ? (odd ? g1 : g0) : my_g;
          const float b0 = group_sum(rgb_part[0][2]), b1 = group_sum(rgb_part[1][2]);
          my_b = mine ? (odd ? b1 : b0) : my_b;
        }
        // PACK: the middle half-step completes ray A (lanes 0..47 of the row), the last one ray B
        const bool of_b = PACK && k == 2;
        const SplitRay fr = of_b ? ray_b : ray;
        const float* tb_f = of_b ? packs + 64 : tb_m;
        if (half == 1 && fr.valid) {
          // ---- lane l holds sample c0 + l: composite the chunk -----------------------------------------------------------
          const float density = expf(my_dlogit) * my_sel;
          const float sem = my_sem + lds[OFF_MISC + 0];
          const float cr = sigmoidf(my_r + lds[OFF_MISC + 1]);
          const float cg = sigmoidf(my_g + lds[OFF_MISC + 2]);
          const float cb = sigmoidf(my_b + lds[OFF_MISC + 3]);
          const int i = c0 + lane;
          const bool valid = i < S;
          const float e0 = tb_f[lane], e1 = tb_f[lane + 1];
          const float mid = (e0 + e1) / 2.f;
          if (PER_SAMPLE) {
            if (valid) {
              const long long o = fr.r * (long long)S + i;
              if (A.s_density) A.s_density[o] = density;
              if (A.s_sem) A.s_sem[o] = sem;
              if (A.s_label) A.s_label[o] = (int64_t)semantics_label(sem);
              if (A.s_rgb) {
                A.s_rgb[3 * o + 0] = cr;
                A.s_rgb[3 * o + 1] = cg;
                A.s_rgb[3 * o + 2] = cb;
              }
              if (A.s_pos) {
                A.s_pos[3 * o + 0] = fr.ox + fr.dx * mid;
                A.s_pos[3 * o + 1] = fr.oy + fr.dy * mid;
                A.s_pos[3 * o + 2] = fr.oz + fr.dz * mid;
              }
            }
          } else {
          const float w = composite_chunk(st, valid, i == S - 1, e1 - e0, density, mid, cr, cg, cb, sem, A.eval_clamp != 0);
          if (A.out_w && valid) A.out_w[fr.r * (long long)S + i] = w;
          bool finish = PACK || k == nhalf - 1;
          if (!finish && A.early_stop > 0.f && __expf(-st.carry_dd) < A.early_stop) {  // wave-uniform
            // the samples behind this chunk carry less than the threshold in total weight: drop them (same rule and same
            // "last sample" stand-in as render_fused_kernel)
            st.last_r = wave_read(A.eval_clamp ? nan_to_num(cr) : cr, 63);
            st.last_g = wave_read(A.eval_clamp ? nan_to_num(cg) : cg, 63);
            st.last_b = wave_read(A.eval_clamp ? nan_to_num(cb) : cb, 63);
            st.last_mid = wave_read(mid, 63);
            if (A.out_w)
              for (int kk = c0 + 64 + lane; kk < S; kk += 64) A.out_w[fr.r * (long long)S + kk] = 0.f;
            if (lane == 0) reinterpret_cast<volatile int*>(ring + PAIR_FLAGS)[2] = (int)qi;
            ray_stopped = true;
            finish = true;
          }
          if (finish) {
            const CompositeOut o = composite_finish(st, A.bg_mode, A.bg[0], A.bg[1], A.bg[2], A.eval_clamp != 0);
            if (lane == 0) {
              const long long r = fr.r;
              if (A.out_acc) A.out_acc[r] = o.acc;
              if (A.out_depth) A.out_depth[r] = o.depth;
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
          }  // !PER_SAMPLE
        }
        if constexpr (PACK) {
          if (k == 1) {
            // ray A's row is done: tile B0 sits in lane row 3 (lanes 48..63) and starts ray B's row at lanes 0..15
            const int src = (lane + 48) & 63;
            my_dlogit = __shfl(my_dlogit, src);
            my_sel = __shfl(my_sel, src);
            my_sem = __shfl(my_sem, src);
            my_r = __shfl(my_r, src);
            my_g = __shfl(my_g, src);
            my_b = __shfl(my_b, src);
            st = CompositeState();
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace cn

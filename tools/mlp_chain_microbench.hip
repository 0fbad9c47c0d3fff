// What would a split-bf16 matrix path buy the render kernels?  (gfx950)
// hipcc -O3 --offload-arch=gfx950 tools/mlp_chain_microbench.hip -o /tmp/mlp_chain && /tmp/mlp_chain
// A chain of NL 64->64 ReLU layers on a 16-sample tile per wave, weights read from LDS for every MFMA, the accumulators
// of one layer handed to the next as its B operand without leaving the lane (weight rows stored in the permuted order
// that makes this work) -- the structure of the field MLPs in render_fused.hip -- in two arithmetics:
//   fp32   v_mfma_f32_16x16x4_f32, 64 per layer (what the kernels do today: exact fp32 products)
//   bf16x3 every operand split into bf16 hi + lo; a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_16x16x32_bf16,
//          fp32 accumulation: 24 MFMAs per layer + the split of the 16 accumulators of a lane
// Reports time per layer-tile and the error of both against an fp64 evaluation of the same chain.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NL = 6;        // layers in the chain
constexpr int NT = 512;      // 8 waves = 2 per SIMD, as the matrix waves of render_split_kernel
// LDS images per layer: fp32 [t(4)][step(16)][lane(64)] floats; bf16 [t(4)][kb(2)][hi/lo][lane(64)][8] bf16
constexpr int W32_PER_LAYER = 4 * 16 * 64;
constexpr int W16_PER_LAYER = 4 * 2 * 2 * 64 * 8;

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
  return v;
}

__global__ void __launch_bounds__(NT) chain_fp32(const float* __restrict__ wimg, const float* __restrict__ x0, float* __restrict__ out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < NL * W32_PER_LAYER; i += NT) lds[i] = wimg[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 act[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 4; ++r) act[t][r] = x0[(16 * t + 4 * (lane >> 4) + r) * 16 + (lane & 15)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
      const float* w = lds + l * W32_PER_LAYER;
      f32x4 nxt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(t * 16 + s) * 64 + lane], act[s >> 2][s & 3], acc, 0, 0, 0);
        nxt[t] = relu4(acc);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) act[t] = nxt[t];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64)
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 4; ++r) out[(16 * t + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = act[t][r];
}

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? a[j] : b[j - 4];
    const __bf16 h = (__bf16)x;
    hi[j] = h;
    lo[j] = (__bf16)(x - (float)h);
  }
}

__global__ void __launch_bounds__(NT) chain_bf16x3(const __bf16* __restrict__ wimg, const float* __restrict__ x0, float* __restrict__ out, int iters) {
  extern __shared__ float lds[];
  __bf16* wl = reinterpret_cast<__bf16*>(lds);
  for (int i = threadIdx.x; i < NL * W16_PER_LAYER; i += NT) wl[i] = wimg[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 act[4];  // act[t][r]: physical row 16 t + 4 g + r = logical feature 32 (t >> 1) + 8 g + 4 (t & 1) + r
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 4; ++r) act[t][r] = x0[(32 * (t >> 1) + 8 * (lane >> 4) + 4 * (t & 1) + r) * 16 + (lane & 15)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
      const bf16x8* w = reinterpret_cast<const bf16x8*>(wl + (size_t)l * W16_PER_LAYER);
      bf16x8 bh[2], bl[2];
      split8(act[0], act[1], bh[0], bl[0]);
      split8(act[2], act[3], bh[1], bl[1]);
      f32x4 nxt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const bf16x8 ah = w[((t * 2 + kb) * 2 + 0) * 64 + lane], al = w[((t * 2 + kb) * 2 + 1) * 64 + lane];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[kb], acc, 0, 0, 0);
        }
        nxt[t] = relu4(acc);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) act[t] = nxt[t];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64)
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 4; ++r) out[(32 * (t >> 1) + 8 * (lane >> 4) + 4 * (t & 1) + r) * 16 + (lane & 15)] = act[t][r];
}

static float to_bf16_round(float x) {  // round to nearest even, as (__bf16)x does
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r; memcpy(&r, &u, 4);
  return r;
}
static unsigned short bf16_bits(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }

int main() {
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  std::vector<float> W((size_t)NL * 64 * 64), X(64 * 16);
  for (auto& w : W) w = rnd() * 0.25f;
  for (auto& x : X) x = rnd();
  // fp64 reference of ONE pass through the chain
  std::vector<double> a(X.begin(), X.end()), b(64 * 16);
  for (int l = 0; l < NL; ++l) {
    for (int o = 0; o < 64; ++o)
      for (int n = 0; n < 16; ++n) {
        double s = 0;
        for (int k = 0; k < 64; ++k) s += (double)W[((size_t)l * 64 + o) * 64 + k] * a[k * 16 + n];
        b[o * 16 + n] = s > 0 ? s : 0;
      }
    a = b;
  }
  // weight images in MFMA operand order.  A operand: lane = (k group, output row i within the tile).
  std::vector<float> img32((size_t)NL * W32_PER_LAYER);
  std::vector<unsigned short> img16((size_t)NL * W16_PER_LAYER);
  for (int l = 0; l < NL; ++l)
    for (int t = 0; t < 4; ++t)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        // fp32 scheme: physical output row 16 t + 4 g' + r holds logical feature 16 t + 4 r + g' (so that step s = (t', r')
        // of the NEXT layer finds feature 4 s + g in register r' of tile t' of lane group g)
        const int out32 = 16 * t + 4 * (i & 3) + (i >> 2);
        for (int s = 0; s < 16; ++s)
          img32[((size_t)l * 64 + t * 16 + s) * 64 + lane] = W[((size_t)l * 64 + out32) * 64 + 4 * s + g];
        // bf16 scheme: physical output row 16 t + i (i = 4 g' + r) holds logical feature 32 (t >> 1) + 8 g' + 4 (t & 1) + r
        const int out16 = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
        for (int kb = 0; kb < 2; ++kb)
          for (int j = 0; j < 8; ++j) {
            const float w = W[((size_t)l * 64 + out16) * 64 + 32 * kb + 8 * g + j];
            const float hi = to_bf16_round(w), lo = to_bf16_round(w - hi);
            img16[((((size_t)l * 4 + t) * 2 + kb) * 2 + 0) * 512 + lane * 8 + j] = bf16_bits(hi);
            img16[((((size_t)l * 4 + t) * 2 + kb) * 2 + 1) * 512 + lane * 8 + j] = bf16_bits(lo);
          }
      }
  // the fp32 kernel reads x0 as physical rows; permute the input accordingly (physical row 16 t + 4 g + r = logical 16 t + 4 r + g)
  std::vector<float> X32(64 * 16);
  for (int t = 0; t < 4; ++t)
    for (int g = 0; g < 4; ++g)
      for (int r = 0; r < 4; ++r)
        for (int n = 0; n < 16; ++n) X32[(16 * t + 4 * g + r) * 16 + n] = X[(16 * t + 4 * r + g) * 16 + n];
  float *d32, *dx, *dx32, *dout;
  __bf16* d16;
  (void)hipMalloc(&d32, img32.size() * 4); (void)hipMalloc(&d16, img16.size() * 2);
  (void)hipMalloc(&dx, 64 * 16 * 4); (void)hipMalloc(&dx32, 64 * 16 * 4); (void)hipMalloc(&dout, 64 * 16 * 4);
  (void)hipMemcpy(d32, img32.data(), img32.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d16, img16.data(), img16.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dx, X.data(), 64 * 16 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dx32, X32.data(), 64 * 16 * 4, hipMemcpyHostToDevice);
  const size_t lds32 = (size_t)NL * W32_PER_LAYER * 4, lds16 = (size_t)NL * W16_PER_LAYER * 2;
  (void)hipFuncSetAttribute((const void*)chain_fp32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);
  (void)hipFuncSetAttribute((const void*)chain_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
  std::vector<float> got(64 * 16);
  auto err = [&](const char* name, bool permuted32) {
    (void)hipMemcpy(got.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int f = 0; f < 64; ++f)
      for (int n = 0; n < 16; ++n) {
        int row = f;
        if (permuted32) { const int t = f >> 4, r = (f >> 2) & 3, g = f & 3; row = 16 * t + 4 * g + r; }  // logical f lives in physical row
        worst = fmax(worst, fabs((double)got[row * 16 + n] - a[f * 16 + n]));
        scale = fmax(scale, fabs(a[f * 16 + n]));
      }
    printf("%-8s max |error| vs fp64 after %d layers: %.3e (largest output %.3f, relative %.2e)\n", name, NL, worst, scale, worst / scale);
  };
  hipLaunchKernelGGL(chain_fp32, dim3(1), dim3(NT), lds32, 0, d32, dx32, dout, 1);
  (void)hipDeviceSynchronize(); err("fp32", true);
  hipLaunchKernelGGL(chain_bf16x3, dim3(1), dim3(NT), lds16, 0, d16, dx, dout, 1);
  (void)hipDeviceSynchronize(); err("bf16x3", false);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 200, blocks = 256;
  for (int variant = 0; variant < 2; ++variant) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, 0);
      if (variant == 0) hipLaunchKernelGGL(chain_fp32, dim3(blocks), dim3(NT), lds32, 0, d32, dx32, dout, iters);
      else hipLaunchKernelGGL(chain_bf16x3, dim3(blocks), dim3(NT), lds16, 0, d16, dx, dout, iters);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
    }
    const double layer_tiles = (double)blocks * 8 * iters * NL;  // one wave = one 16-sample tile per layer
    printf("%-8s %.3f ms  -> %.1f ns per (64x64 layer, 16 samples) per CU-wave; %.2f TMAC/s of logical work\n",
           variant ? "bf16x3" : "fp32", best, best * 1e6 / layer_tiles * blocks * 8, layer_tiles * 64 * 64 * 16 / best * 1e-9);
  }
  return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../cropnerf-a-neural-radiance-field-based-framework_amd/csrc/cn_common.hpp"
namespace cn { void set_error(const char*, ...) {} }
__global__ void k(const float* tab, unsigned T, int n, const float* pos, float2* a, float2* b) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cn::u32x4_t r = cn::table_rsrc(tab, 4u * T * 8u);
  unsigned lvl = (i % 4) * T;
  float scale = 16.f + 37.f * (i % 4);
  a[i] = cn::hash_level(tab, lvl, T - 1, scale, pos[3*i], pos[3*i+1], pos[3*i+2]);
  cn::XPairLoads l0, l1;
  cn::xpair_issue(l0, r, lvl, T - 1, scale, pos[3*i], pos[3*i+1], pos[3*i+2]);
  cn::xpair_issue(l1, r, lvl, T - 1, scale, pos[3*i], pos[3*i+1], pos[3*i+2]);
  cn::xpair_wait(l0, l1);
  b[i] = cn::xpair_blend(l1, scale, pos[3*i], pos[3*i+1], pos[3*i+2]);
}
int main() {
  unsigned T = 1 << 14; int n = 4096;
  std::vector<float> tab(4 * T * 2), pos(3 * n);
  for (auto& v : tab) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : pos) v = (float)rand() / RAND_MAX;
  float *dt, *dp; float2 *da, *db;
  hipMalloc(&dt, tab.size()*4); hipMalloc(&dp, pos.size()*4); hipMalloc(&da, n*8); hipMalloc(&db, n*8);
  hipMemcpy(dt, tab.data(), tab.size()*4, hipMemcpyHostToDevice); hipMemcpy(dp, pos.data(), pos.size()*4, hipMemcpyHostToDevice);
  k<<<n/256, 256>>>(dt, T, n, dp, da, db);
  std::vector<float2> a(n), b(n);
  hipMemcpy(a.data(), da, n*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n*8, hipMemcpyDeviceToHost);
  int bad = 0, badodd = 0, odd = 0;
  for (int i = 0; i < n; ++i) {
    float scale = 16.f + 37.f * (i % 4);
    int ix = (int)floorf(pos[3*i] * scale);
    odd += ix & 1;
    if (fabsf(a[i].x - b[i].x) > 1e-5f || fabsf(a[i].y - b[i].y) > 1e-5f) { bad++; badodd += ix & 1; if (bad < 6) printf("i=%d ix=%d a=(%f,%f) b=(%f,%f)\n", i, ix, a[i].x, a[i].y, b[i].x, b[i].y); }
  }
  printf("bad %d (of which odd-ix %d) / %d, odd total %d\n", bad, badodd, n, odd);
  return 0;
}

// cn_deterministic_*: the registry and the flush pass of the deterministic-accumulation test build (cn_det.hpp).  In the
// default build the entry points exist and say so: cn_deterministic_build() == 0, register fails with CN_ERR_UNSUPPORTED.
#include "cn_common.hpp"
#include "cn_det.hpp"

#include <vector>

#if CN_DETERMINISTIC_SCATTER
namespace cn {
namespace {
std::mutex g_det_mutex;
DetTable g_det_host{};
std::vector<int (*)(const DetTable*)>& uploaders() {
  static std::vector<int (*)(const DetTable*)> v;
  return v;
}
int upload_all() {
  for (auto fn : uploaders())
    if (int rc = fn(&g_det_host)) {
      set_error("cn_deterministic: copying the range table to a translation unit's device copy failed");
      return rc;
    }
  return 0;
}
__global__ void __launch_bounds__(256) det_flush_kernel(DetTable t) {
  for (int k = 0; k < t.n; ++k) {
    float* dst = t.r[k].base;
    long long* sh = t.r[k].shadow;
    const unsigned long long n = t.r[k].count;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull) {
      const long long s = sh[i];
      if (s != 0) {
        dst[i] += (float)((double)s * (1.0 / DET_SCALE));
        sh[i] = 0;
      }
    }
  }
}
}  // namespace

void det_add_uploader(int (*fn)(const DetTable*)) { uploaders().push_back(fn); }

int det_flush(hipStream_t stream) {
  DetTable t;
  {
    std::lock_guard<std::mutex> lock(g_det_mutex);
    t = g_det_host;
  }
  if (t.n == 0) return CN_OK;
  unsigned long long most = 0;
  for (int k = 0; k < t.n; ++k) most = t.r[k].count > most ? t.r[k].count : most;
  hipLaunchKernelGGL(det_flush_kernel, dim3(grid_for((long long)most, 256, 8192)), dim3(256), 0, stream, t);
  return check_launch("cn_deterministic_flush");
}
}  // namespace cn
#endif

extern "C" int cn_deterministic_build(void) { return CN_DETERMINISTIC_SCATTER ? 1 : 0; }

extern "C" int cn_deterministic_register(float* base, int64_t count, int64_t* shadow, uint64_t* miss_counter) {
#if CN_DETERMINISTIC_SCATTER
  CN_REQUIRE(base && shadow && count > 0, CN_ERR_INVALID, "cn_deterministic_register: null range or count <= 0");
  std::lock_guard<std::mutex> lock(cn::g_det_mutex);
  CN_REQUIRE(cn::g_det_host.n < cn::DET_MAX_RANGES, CN_ERR_INVALID, "cn_deterministic_register: more than %d ranges",
             cn::DET_MAX_RANGES);
  for (int k = 0; k < cn::g_det_host.n; ++k) {
    const cn::DetRange& r = cn::g_det_host.r[k];
    CN_REQUIRE(base + count <= r.base || r.base + r.count <= base, CN_ERR_INVALID,
               "cn_deterministic_register: the range overlaps a registered one");
  }
  cn::g_det_host.r[cn::g_det_host.n++] = cn::DetRange{base, (unsigned long long)count, reinterpret_cast<long long*>(shadow)};
  if (miss_counter) cn::g_det_host.misses = reinterpret_cast<unsigned long long*>(miss_counter);
  return cn::upload_all();
#else
  (void)base, (void)count, (void)shadow, (void)miss_counter;
  cn::set_error("cn_deterministic_register: this is the default build (float atomics); the deterministic accumulation mode "
                "is libcropnerf_hip_det.so (built beside the default library by build.py; CN_DETERMINISTIC_SCATTER=1 selects it in cropnerf_amd/_lib.py)");
  return CN_ERR_UNSUPPORTED;
#endif
}

extern "C" int cn_deterministic_clear(void) {
#if CN_DETERMINISTIC_SCATTER
  std::lock_guard<std::mutex> lock(cn::g_det_mutex);
  cn::g_det_host = cn::DetTable{};
  return cn::upload_all();
#else
  return CN_OK;
#endif
}

extern "C" int cn_deterministic_flush(cn_stream_t stream) {
#if CN_DETERMINISTIC_SCATTER
  return cn::det_flush(cn::as_stream(stream));
#else
  (void)stream;
  return CN_OK;
#endif
}

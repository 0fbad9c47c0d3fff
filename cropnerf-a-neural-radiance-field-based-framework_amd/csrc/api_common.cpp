// Error reporting and version for libcropnerf_hip.so.
#include "cn_common.hpp"

namespace cn {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cn

extern "C" const char* cn_last_error(void) { return cn::g_err; }
extern "C" int cn_version(void) { return 100; }

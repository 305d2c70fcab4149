// Error reporting + small global knobs of the C ABI (include/agcn.h).
#include "agcn_internal.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
std::atomic<int32_t> g_lds_limit{160 * 1024};
}  // namespace

namespace agcn {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace agcn

extern "C" const char* agcn_last_error(void) { return g_err; }
extern "C" int32_t agcn_version(void) { return AGCN_VERSION; }
extern "C" int32_t agcn_get_lds_limit(void) { return g_lds_limit.load(); }
extern "C" void agcn_set_lds_limit(int32_t bytes) {
  if (bytes < 0) bytes = 0;
  if (bytes > 160 * 1024) bytes = 160 * 1024;
  g_lds_limit.store(bytes);
}

// Library-level entry points: version and the thread-local error string of the C ABI.
#include "common.h"

namespace ucd {
namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
  return (int)e;
}
}  // namespace ucd

extern "C" {
int ucd_version(void) { return UCD_VERSION; }
const char* ucd_last_error(void) { return ucd::g_error; }

int ucd_fill_zero(void* ptr, size_t bytes, ucd_stream_t stream) {
  if (!ptr || bytes == 0) return 0;
  hipError_t e = hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream);
  if (e == hipSuccess) return 0;
  (void)hipGetLastError();
  ucd::set_error("ucd_fill_zero: hipMemsetAsync: %s", hipGetErrorString(e));
  return (int)e;
}
}

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
}

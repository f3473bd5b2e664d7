// Error plumbing and library-level entry points of the C-ABI (include/dskd_hip.h).
#include "common.h"
#include <string.h>

namespace dskd {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace dskd

extern "C" int dskd_abi_version(void) { return 2; }

extern "C" const char* dskd_last_error(void) { return dskd::err_buf(); }

extern "C" int dskd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

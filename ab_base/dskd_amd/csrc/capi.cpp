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

extern "C" int dskd_zero_fill(void* p, int64_t bytes, void* stream) {
  if (bytes < 0 || (bytes > 0 && !p) || (bytes & 15) || (reinterpret_cast<uintptr_t>(p) & 15))
    return dskd::fail(DSKD_ERR_INVALID_ARG, "dskd_zero_fill: need a 16-byte aligned buffer of a multiple of 16 bytes");
  dskd::zero_fill(p, (size_t)bytes, (hipStream_t)stream);
  return dskd::check_launch("dskd_zero_fill");
}

extern "C" int dskd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

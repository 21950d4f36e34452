// common.hip -- error reporting and device queries of the C ABI.
#include <cstring>
#include "sntc_internal.h"

namespace sntc {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  g_last_error = std::string("HIP error in ") + what + ": " + hipGetErrorString(e);
  (void)hipGetLastError();
  return SNTC_ERR_HIP;
}

}  // namespace sntc

extern "C" const char* sntc_last_error(void) { return sntc::g_last_error.c_str(); }

extern "C" int sntc_version(void) { return SNTC_VERSION; }

extern "C" int sntc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int sntc_device_arch(int device, char* buf, size_t buflen) {
  if (!buf || buflen == 0) return sntc::fail(SNTC_ERR_BAD_SHAPE, "sntc_device_arch: null buffer");
  hipDeviceProp_t prop;
  if (sntc_device_count() <= device || device < 0) return sntc::fail(SNTC_ERR_NO_DEVICE, "no such HIP device");
  SNTC_HIP(hipGetDeviceProperties(&prop, device));
  std::strncpy(buf, prop.gcnArchName, buflen - 1);
  buf[buflen - 1] = 0;
  return SNTC_OK;
}

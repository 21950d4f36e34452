// common.hip -- error reporting and device queries of the C ABI.
#include <algorithm>
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

// Zero `bytes` (a multiple of 4, 4-byte aligned) on `stream` with a KERNEL, not hipMemsetAsync: the accumulators, hand-off flags and
// work queues of this library are re-armed on the launch stream in front of every launch that uses them, and a memset recorded
// into a HIP graph through torch's stream capture was found not to re-arm them on replay (stream-K results wrong from the second
// replay on, tools/graph_decode_check.py); a kernel node replays like every other launch.
__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

int zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return SNTC_OK;
  if ((bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return fail(SNTC_ERR_BAD_SHAPE, "zero_async: unaligned range");
  const size_t n = bytes / 4;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(p), n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "zero_async");
  return SNTC_OK;
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

// device_math.h -- device helpers shared by the entropy / SGA / training translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLogScaleMin = -2.2072749131897207f;    // ln 0.11                  (mshyper/models.py:29)
constexpr float kScaleFactor = 0.12305479932808384f;    // (ln 256 - ln 0.11) / 63  (:31)
constexpr float kInvLn2 = 1.4426950408889634f;

// log Phi(x), float32: direct for x > -10, asymptotic series below (as TFP's float32 log_ndtr).
__device__ __forceinline__ float log_ndtr_f(float x) {
  const float t = x * 0.70710678118654752f;
  if (x > 0.0f) return log1pf(-0.5f * erfcf(t));
  if (x > -10.0f) return logf(0.5f * erfcf(-t));
  const float x2 = x * x;
  const float ix2 = 1.0f / x2;
  const float series = 1.0f - ix2 * (1.0f - 3.0f * ix2 * (1.0f - 5.0f * ix2));
  return -0.5f * x2 - logf(-x) - 0.91893853320467274f + logf(series);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float log_sigmoid_f(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

// counter-based generator: one 64-bit hash per (seed, step, element)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned long long stream_key(unsigned long long seed, unsigned long long step) {
  return splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull));
}
__device__ __forceinline__ float uniform01(unsigned long long r) {            // (0, 1)
  return ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

// block-wide sum (blockDim.x a multiple of 64, <= 512) added to *dst with one double atomic
__device__ __forceinline__ void block_sum_to(double v, double* dst) {
  __shared__ double part[8];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += part[i];
    atomicAdd(dst, s);
  }
  __syncthreads();
}

}  // namespace sntc

// Sustained fp32-MFMA rate of the whole chip: every wave issues independent v_mfma_f32_32x32x2_f32 back to back
// (no memory traffic), timed with HIP events.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a, float b) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd;     // 256 threads = 4 waves = one per SIMD
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = rep == 0 ? 2000 : 40000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
      if (rep) printf("waves/SIMD %d: %.2f ms  %.1f TFLOP/s  (%.0f%% of 157.3)\n", waves_per_simd, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    }
  }
  return 0;
}

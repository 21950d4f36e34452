// Do the fp32 matrix pipe and the fp32 vector pipe of a CU add up?  512-thread workgroups (two waves per SIMD), one per CU;
// waves 0-3 issue independent v_mfma_f32_32x32x2_f32 back to back, waves 4-7 issue independent v_pk_fma_f32 (or v_fma_f32) on
// registers; modes: matrix only (both halves MFMA / one half idle), vector only, mixed.  Random-ish operands (the clock the
// chip holds depends on the data).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bits: 1 = waves 0-3 run MFMA, 2 = waves 4-7 run MFMA, 4 = waves 4-7 run packed FMA, 8 = waves 0-3 run packed FMA
__global__ void __launch_bounds__(512, 2) mix_loop(float* out, int iters, int mode, const float* seed) {
  const int wave = threadIdx.x >> 6;
  const bool lo = wave < 4;
  const float a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
  const bool do_mfma = lo ? (mode & 1) : (mode & 2);
  const bool do_fma = lo ? (mode & 8) : (mode & 4);
  float s = 0;
  if (do_mfma) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  } else if (do_fma) {
    // 4 MFMAs = 256 cycles of the matrix pipe per iteration; the vector half does 64 packed FMAs (2 cycles each at full rate: 128)
    f32x2 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f32x2{a + k, b - k};
    const f32x2 x = {a, b}, y = {b * 0.999f, a * 1.001f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_elementwise_fma(acc[k], x, y);
    }
    for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1];
  }
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float *d, *seed;
  hipMalloc(&d, 4);
  hipMalloc(&seed, 128 * 4);
  float h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.5f + 0.37f * (float)((i * 2654435761u >> 7) & 255) / 256.0f;
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"matrix on 4 waves per CU (one per SIMD), other 4 idle", "matrix on all 8 waves", "packed vector FMA on 4 waves, other 4 idle",
                         "packed vector FMA on all 8 waves", "MIXED: matrix on waves 0-3, packed vector FMA on waves 4-7"};
  const int modes[] = {1, 3, 4, 12, 5};
  for (int m = 0; m < 5; ++m) {
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = rep == 0 ? 2000 : 30000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(mix_loop, dim3(256), dim3(512), 0, 0, d, iters, modes[m], seed);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const int mw = ((modes[m] & 1) ? 4 : 0) + ((modes[m] & 2) ? 4 : 0), vw = ((modes[m] & 4) ? 4 : 0) + ((modes[m] & 8) ? 4 : 0);
      const double fm = 256.0 * mw * iters * 4 * (2.0 * 32 * 32 * 2), fv = 256.0 * vw * iters * 64 * (64 * 2 * 2.0);
      if (rep == 2)
        printf("%-62s %.2f ms  matrix %.1f + vector %.1f = %.1f TFLOP/s\n", names[m], ms, fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
    }
  }
  return 0;
}

// rb_common.h -- what the two instantiations of the whole-ResidualBlock kernel share (rb_fused.hip: exact fp32;
// rb_fused_bf3.hip: bf16 x 3 split precision): tile geometry, kernel arguments, small device helpers.
#pragma once
#include <type_traits>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace rb {

constexpr unsigned kOOB = 0x80000000u;   // beyond any buffer (< 2 GiB, host check): loads give zeros, stores are dropped

template <int C>
struct RBCfg {
  static constexpr int CH = C / 2;             // hidden channels
  static constexpr int NT = CH / 32;           // 32-channel tiles of the hidden width
  static constexpr int SL = CH / 16;           // 16-channel slabs of the hidden width
  static constexpr int TH = 8, TW = 32;        // output tile: rows (= waves) x pixels (= one MFMA fragment)
  static constexpr int PW = TW + 2, PH = TH + 2, PP = PW * PH;   // halo patch
  static constexpr int NPT = (PP + 31) / 32;   // 32-pixel tiles of the patch the head computes
  static constexpr int UNIT = CH * 16;         // floats per ring unit
  static constexpr int U0 = C / 16;            // head units: K stages of the c -> c/2 contraction
  static constexpr int U1 = SL * 9;            // 3x3 units: (slab, tap)
  static constexpr int U2 = (C / 32) * 2;      // tail units: (32-channel output tile, half of K = c/2)
  static constexpr int UT = U0 + U1 + U2;
  static constexpr int RING = 3;
  static constexpr int PATCH = SL * PP * 16;   // floats
  static constexpr int BIAS = CH + CH + C;     // floats: b0 | b1 | b2
  static constexpr size_t LDS = (size_t)(PATCH + RING * UNIT + BIAS) * 4;
  static_assert(UT % RING == 0 && U0 % RING == 0 && (U0 + U1) % RING == 0, "ring slots are compile-time per step");
  static_assert((UNIT * 4) % 1024 == 0, "a unit is a whole number of 1-KB LDS-DMA pieces (64 lanes x 16 B)");
  static_assert(NPT > 8 && NPT <= 16, "head: every wave one patch tile, the first NPT - 8 waves two");
};

struct RBArgs {
  const float* x;
  float* y;
  const float* wpack;      // [UT][CH][16] ring units, LDS image order (swizzled)
  const float* bias;       // [CH + CH + C]: b0 | b1 | b2 (zeros where a layer has none)
  unsigned bytes;          // size of x and of y
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t rsrc, f32x4 v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)voff, (int)soff, 0);
}

// 4 x 4 transpose of 16-B quads across the four lanes of a lane quad (lanes 4 m .. 4 m + 3), in registers (two rounds of DPP
// quad_perm exchanges): T[j] of lane i <- T[i] of lane j.  The tail's accumulators hold, per lane = pixel, the register quad q
// = channels 8 q + 4 h .. + 3; after the transpose lane i of a quad holds chunk 2 i + h of the FOUR pixels of its quad, so that
// store (and residual load) j of the output tile covers whole 128-B lines of eight pixels -- eight consecutive lanes write one
// pixel's 32 channels -- instead of 32-B pieces of 32 lines: the texture addresser, not the MFMA pipe, was timing the tail.
__device__ __forceinline__ float rb_dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float rb_dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void quad_transpose(f32x4 (&T)[4], int lane_in_quad) {
  const bool o1 = (lane_in_quad & 1) != 0, o2 = (lane_in_quad & 2) != 0;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float got = rb_dpp_xor1(o1 ? T[2 * b][e] : T[2 * b + 1][e]);
      T[2 * b][e] = o1 ? got : T[2 * b][e];
      T[2 * b + 1][e] = o1 ? T[2 * b + 1][e] : got;
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float got = rb_dpp_xor2(o2 ? T[q][e] : T[q + 2][e]);
      T[q][e] = o2 ? got : T[q][e];
      T[q + 2][e] = o2 ? T[q + 2][e] : got;
    }
}

}  // namespace rb

// bf16 x 3 instantiation (rb_fused_bf3.hip)
int rb3_init();
size_t rb3_pack_bytes();                                    // bytes of the packed split-precision weight stream (c = 192)
int rb3_pack(const float* w0, const float* w1, const float* w2, void* wpack3, hipStream_t s);
int rb3_launch(const rb::RBArgs& a, int grid, hipStream_t s);

}  // namespace sntc

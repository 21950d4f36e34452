// rb_fused_bf3.hip -- the whole-ResidualBlock kernel of rb_fused.hip (reference common/elic.py:41-68) in SPLIT PRECISION:
// every fp32 operand as three bfloat16 terms hi + mid + lo (24 mantissa bits), a product from its six significant cross
// terms (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, smallest first) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// 6 MFMAs of 32 cycles per 16-deep stage and 32 x 32 tile against 8 fp32 MFMAs of 64.  Same accuracy class
// as fp32 against float64, NOT bit-identical to it: Model(precision="bf16x3") only.
//
// Same structure as rb_fused.hip -- 8 x 32 pixel tile, head on the halo patch into LDS (the patch stays fp32: in the split
// format it would not fit), 3x3 as nine shifted reads of the patch, accumulators as the tail's operand, tiles walked down
// the image with the patch as a ring of rows, one LDS-DMA weight stream -- with these differences:
//   * a unit of the weight stream is PRE-SPLIT at plan creation: [plane hi | mid | lo][96 rows][16 bf16 = 32 B], the two 16-B
//     halves of a row swapped on rows 8-15 (mod 16) so that the fragment ds_read_b128 is conflict-free; 9 KB, nine 1-KB pieces;
//   * a 32x32x16 operand is eight CONSECUTIVE k per lane (k = 8 h + j): pixel fragments are two 16-B reads (chunks 2 h, 2 h + 1
//     of the 64-B patch row / two 16-B global loads in the head) split into the three planes in registers -- ~44 vector
//     instructions per stage, which issue under the stage's 18 MFMAs;
//   * the tail's operand is the accumulator tile converted once: registers 8 s .. 8 s + 7 of a channel tile are k-step s, element
//     j of lane half h being channel 16 s + 8 (j >> 2) + 4 h + (j & 3) -- W2 is packed in that k order.
#include <algorithm>
#include <mutex>
#include <type_traits>
#include "rb_common.h"

namespace sntc {
using namespace rb;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kUnit3 = 3 * 96 * 32;            // bytes of one pre-split unit (c = 192): 3 planes x 96 rows x 16 bf16
constexpr int kRing3 = 3;

struct S3 {
  bf16x8 p[3];                                  // hi, mid, lo
};

__device__ __forceinline__ void split1(float x, __bf16* hi, __bf16* mid, __bf16* lo) {
  const __bf16 hh = (__bf16)x;
  const float r1 = x - (float)hh;               // exact
  const __bf16 mm = (__bf16)r1;
  const float r2 = r1 - (float)mm;              // exact
  *hi = hh; *mid = mm; *lo = (__bf16)r2;
}

// Eight fp32 values -> three planes of eight bfloat16, by TRUNCATION: hi = the top 16 bits of x, mid = the top 16 bits of
// x - hi, lo = the top 16 bits of x - hi - mid.  Every term takes the 8 leading significant bits of what is left, so
// hi + mid + lo == x exactly (24 significant bits), as with the round-to-nearest split of the weights; per pair of values:
// 2 AND, 2 SUB, 2 AND, 2 SUB and three byte permutes that put the two values' upper halves side by side -- 44 plain vector
// instructions per fragment (the cast-based form compiled to ~65, among them v_pk_add_f32, which is expensive next to MFMAs).
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ S3 split8(const f32x4 a, const f32x4 b) {
  u32x4v P[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = q < 2 ? a[2 * q] : b[2 * q - 4], x1 = q < 2 ? a[2 * q + 1] : b[2 * q - 3];
    const unsigned h0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u, h1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
    const float r0 = x0 - __builtin_bit_cast(float, h0), r1 = x1 - __builtin_bit_cast(float, h1);
    const unsigned m0 = __builtin_bit_cast(unsigned, r0) & 0xffff0000u, m1 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float s0 = r0 - __builtin_bit_cast(float, m0), s1 = r1 - __builtin_bit_cast(float, m1);
    // v_perm_b32: bytes 0-3 of the selector's value space are the SECOND operand, 4-7 the first: [x1.hi16 : x0.hi16]
    P[0][q] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    P[1][q] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    P[2][q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
  }
  S3 o;
#pragma unroll
  for (int p = 0; p < 3; ++p) o.p[p] = __builtin_bit_cast(bf16x8, P[p]);
  return o;
}

#define RB3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// acc += W (pre-split planes w[0..2]) * X (split planes x.p[0..2]); weights are the A operand (rows = channels)
__device__ __forceinline__ f32x16 mac6(const bf16x8 (&w)[3], const S3& x, f32x16 acc) {
  acc = RB3_MFMA(w[2], x.p[0], acc);             // lo * hi
  acc = RB3_MFMA(w[0], x.p[2], acc);             // hi * lo
  acc = RB3_MFMA(w[1], x.p[1], acc);             // mid * mid
  acc = RB3_MFMA(w[1], x.p[0], acc);             // mid * hi
  acc = RB3_MFMA(w[0], x.p[1], acc);             // hi * mid
  acc = RB3_MFMA(w[0], x.p[0], acc);             // hi * hi
  return acc;
}

// term t (0 .. 5, smallest first) of acc += W * X
__device__ __forceinline__ f32x16 mac1(const bf16x8 (&w)[3], const S3& x, f32x16 acc, int t) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  return RB3_MFMA(w[PA[t]], x.p[PB[t]], acc);
}

// three accumulators (the three channel tiles of one pixel fragment), their chains INTERLEAVED: row tile 0's fragments are in
// registers when the step begins, tiles 1 and 2 arrive from LDS under its first three MFMAs; after that no MFMA follows one on
// the same accumulator directly
__device__ __forceinline__ void mac6x3(const bf16x8 (&w0)[3], const bf16x8 (&w1)[3], const bf16x8 (&w2)[3], const S3& x, f32x16& a0,
                                       f32x16& a1, f32x16& a2) {
  a0 = mac1(w0, x, a0, 0);
  a0 = mac1(w0, x, a0, 1);
  a0 = mac1(w0, x, a0, 2);
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    a1 = mac1(w1, x, a1, t);
    a2 = mac1(w2, x, a2, t);
    if (t < 3) a0 = mac1(w0, x, a0, t + 3);
  }
}

template <int C>
__global__ void __launch_bounds__(512, 2) rb3_kernel(const RBArgs a) {
  using K = RBCfg<C>;
  constexpr int CH = K::CH, NT = K::NT, SL = K::SL, PW = K::PW, PP = K::PP, PH = K::PH, NEW = K::TH * K::PW;
  constexpr int U0 = K::U0, U1 = K::U1, UT = K::UT, RING = kRing3, UNITB = 3 * CH * 32;
  static_assert(UNITB == kUnit3 && UT % RING == 0 && U0 % RING == 0 && (U0 + U1) % RING == 0, "ring slots are compile-time per step");
  typedef __attribute__((address_space(3))) void lds_void;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);                    // [SL][PP][16] fp32, 16-B chunks XOR-swizzled by (pixel >> 2) & 3
  char* ring = smem + (size_t)K::PATCH * 4;                         // [RING][3 planes][CH rows][32 B]
  float* lbias = reinterpret_cast<float*>(ring + RING * UNITB);     // b0 | b1 | b2

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int G = gridDim.x, b = blockIdx.x;
  const int wl = (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int t_lo = (int)((long long)a.ntiles * wl / G);
  const int t_hi = (int)((long long)a.ntiles * (wl + 1) / G);
  if (t_lo >= t_hi) return;

  const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ws =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, UT * UNITB, 0x00020000);

  // weight fragments of a unit: plane p, row tile j: 16 B at row 32 j + l31, half h ^ ((row >> 3) & 1): k = 8 h + (0..7)
  struct WF { bf16x8 w[NT][3]; };
  const int wfoff = l31 * 32 + ((h ^ ((l31 >> 3) & 1)) << 4);
  // Row tile 0 of a unit is fetched a step ahead (under the previous unit's MFMAs), row tiles 1 .. NT - 1 at the step's start
  // (the step's first 6 MFMAs cover them): 48 fragment registers instead of the 72 a whole-unit double buffer takes
  auto read_rest = [&](WF& F, int slot) {
    const char* base = ring + slot * UNITB + wfoff;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 1; j < NT; ++j) F.w[j][p] = *reinterpret_cast<const bf16x8*>(base + p * (CH * 32) + j * 32 * 32);
  };
  auto read_w1 = [&](bf16x8 (&F)[3], int slot, int rowblock) {     // one row tile (runtime index): the head's extra unit
    const char* base = ring + slot * UNITB + rowblock * 32 * 32 + wfoff;
#pragma unroll
    for (int p = 0; p < 3; ++p) F[p] = *reinterpret_cast<const bf16x8*>(base + p * (CH * 32));
  };

  // unit -> ring slot: nine 1-KB pieces, piece i by wave i % 8
  const unsigned dma_voff = (unsigned)lane * 16u;
  auto dma = [&](int unit, int slot) {
#pragma unroll
    for (int i = 0; i < UNITB / 1024; i += 8) {
      if (i + wave < UNITB / 1024) {
        char* dst = ring + slot * UNITB + (i + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)dst, 16, (int)dma_voff, unit * UNITB + (i + wave) * 1024, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sync = [&](auto VM) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(decltype(VM)::value) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using VM0 = std::integral_constant<int, 0>;
  using VM4 = std::integral_constant<int, 4>;

  for (int i = tid; i < K::BIAS; i += 512) lbias[i] = a.bias[i];
  dma(0, 0);
  dma(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  WF Wc;
  bf16x8 Wn0[3];
  read_w1(Wc.w[0], 0, 0);

  // head geometry (rb_fused.hip): fragment tiles of 32 flat patch pixels of the rows computed
  const bool two = wave < K::NPT - 8;
  static_assert(K::NPT - 8 == NT, "the same waves take the second tile (full) and one channel tile of tile 8 (incremental)");
  int hf[3], hfy[3], hfx[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    hf[p] = 32 * (p == 0 ? wave : p == 1 ? wave + 8 : 8) + l31;
    hfy[p] = hf[p] / PW;
    hfx[p] = hf[p] - hfy[p] * PW;
  }
  unsigned hv[2];
  bool hvalid[2], hin[2];
  int hpp[2];
  bool h_inc = false;
  f32x4 X[3][2][2];                            // [K stage % 3][patch tile][16-B half of the lane's 8 channels]
  int n = 0, y0 = 0, x0 = 0;
  auto coords = [&](int tile, int* tn, int* ty0, int* tx0, bool* inc) {
    const int per = a.tiles_x * a.tiles_y;
    *tn = tile / per;
    const int r = tile - *tn * per;
    const int txi = r / a.tiles_y;
    const int tyi = r - txi * a.tiles_y;
    *ty0 = tyi * K::TH;
    *tx0 = txi * K::TW;
    *inc = tile > t_lo && tyi > 0;
  };
  auto head_setup = [&](int tile) {
    int tn, ty0, tx0;
    coords(tile, &tn, &ty0, &tx0, &h_inc);
    const int r0 = h_inc ? 2 : 0;
    const int ybase = ty0 % PH + r0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int q = p == 0 ? 0 : (h_inc ? 2 : 1);
      const int fy = hfy[q], fx = hfx[q];
      const int iy = ty0 - 1 + r0 + fy, ix = tx0 - 1 + fx;
      int pr = ybase + fy;
      pr = pr >= PH ? pr - PH : pr;
      pr = pr >= PH ? pr - PH : pr;
      hin[p] = hf[q] < (h_inc ? NEW : PP);
      hvalid[p] = hin[p] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      hv[p] = hvalid[p] ? ((unsigned)((tn * a.H + iy) * a.W + ix) * (unsigned)(C * 4) + (unsigned)h * 32u) : kOOB;
      hpp[p] = pr * PW + fx;
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      X[st][0][0] = buf_load(xs, hv[0], st * 64);
      X[st][0][1] = buf_load(xs, hv[0], st * 64 + 16);
      if (two) {
        X[st][1][0] = buf_load(xs, hv[1], st * 64);
        X[st][1][1] = buf_load(xs, hv[1], st * 64 + 16);
      }
    }
  };
  head_setup(t_lo);

#ifdef SNTC_DIAG
  // make DIAG=1: cycles per phase (s_memtime around head / 3x3 / tail, summed over the workgroup's tiles) overwrite the first
  // floats of y at the end of the launch -- tools/rb_phases.py reads them; results are meaningless in such a build
  unsigned long long tph[4] = {0, 0, 0, 0};
#define RB3_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define RB3_STAMP(v)
#endif
  for (int tile = t_lo; tile < t_hi; ++tile) {
    bool inc_now;
    coords(tile, &n, &y0, &x0, &inc_now);
    RB3_STAMP(ts0);

    // ================================================================================================
    // head
    // ================================================================================================
    auto head = [&](auto NPXc, auto EXc) {
      constexpr int NPX = decltype(NPXc)::value;
      constexpr bool EX = decltype(EXc)::value;
      constexpr int NLD = NPX + (EX ? 1 : 0);
      f32x16 acc[NT][NPX], accx;
#pragma unroll
      for (int e = 0; e < 16; ++e) accx[e] = 0.0f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < NPX; ++p)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[j][p][e] = 0.0f;
      // the split of a stage's pixel fragments (~65 vector instructions per fragment tile) runs a step ahead, between the previous
      // stage's MFMAs: a stage that begins with its own split leaves the matrix pipe idle for the ~300 cycles the chain takes
      S3 xs3[NLD];
#pragma unroll
      for (int p = 0; p < NLD; ++p) xs3[p] = split8(X[0][p][0], X[0][p][1]);
      static_for<0, U0>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr bool last = j == U0 - 1;
        constexpr int AHEAD = NPX == 2 ? 1 : 2;
        constexpr bool ld = j + AHEAD < U0 && j + AHEAD >= 2;
        dma(j + 2, (j + 2) % RING);
        if constexpr (ld) {
#pragma unroll
          for (int p = 0; p < NLD; ++p) {
            X[(j + AHEAD) % 3][p][0] = buf_load(xs, hv[p], (j + AHEAD) * 64);
            X[(j + AHEAD) % 3][p][1] = buf_load(xs, hv[p], (j + AHEAD) * 64 + 16);
          }
        }
        read_rest(Wc, j % RING);
        read_w1(Wn0, (j + 1) % RING, 0);         // the next unit's first row tile travels under this unit's MFMAs
        bf16x8 Fx[3];
        if constexpr (EX) read_w1(Fx, j % RING, wave);
#pragma unroll
        for (int p = 0; p < NPX; ++p) mac6x3(Wc.w[0], Wc.w[1], Wc.w[2], xs3[p], acc[0][p], acc[1][p], acc[2][p]);
        if constexpr (EX) accx = mac6(Fx, xs3[NPX], accx);
        S3 xn3[NLD];
        if constexpr (!last) {
#pragma unroll
          for (int p = 0; p < NLD; ++p) xn3[p] = split8(X[(j + 1) % 3][p][0], X[(j + 1) % 3][p][1]);
        }
        // the fragment reads behind the first MFMA, then the next stage's split dealt between the MFMAs (mask 0x8 MFMA, 0x100 DS
        // read, 0x2 VALU)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NT + (EX ? 3 : 0), 0);
        if constexpr (!last) {
#pragma unroll
          for (int q = 0; q < 6 * NT * NPX + (EX ? 6 : 0) - 1; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (72 * NLD + 6 * NT * NPX) / (6 * NT * NPX), 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!last) {
#pragma unroll
          for (int p = 0; p < NLD; ++p) xs3[p] = xn3[p];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) Wc.w[0][p] = Wn0[p];
        if constexpr (last) {
#pragma unroll
          for (int p = 0; p < NPX; ++p) {
            const int sw = (hpp[p] >> 2) & 3;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + 32 * jt + 8 * q + 4 * h);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = hvalid[p] ? fmaxf(acc[jt][p][4 * q + e] + bv[e], 0.0f) : 0.0f;
                if (hin[p])
                  *reinterpret_cast<f32x4*>(patch + (2 * jt + (q >> 1)) * (PP * 16) + hpp[p] * 16 + (((2 * (q & 1) + h) ^ sw) << 2)) = v;
              }
          }
          if constexpr (EX) {
            const int sw = (hpp[NPX] >> 2) & 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + 32 * wave + 8 * q + 4 * h);
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = hvalid[NPX] ? fmaxf(accx[4 * q + e] + bv[e], 0.0f) : 0.0f;
              if (hin[NPX])
                *reinterpret_cast<f32x4*>(patch + (2 * wave + (q >> 1)) * (PP * 16) + hpp[NPX] * 16 + (((2 * (q & 1) + h) ^ sw) << 2)) = v;
            }
          }
        }
        if constexpr (ld && AHEAD == 2) sync(std::integral_constant<int, 2 * NLD>{});
        else sync(VM0{});
      });
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    if (!two) head(I1{}, No{});
    else if (inc_now) head(I1{}, Yes{});
    else head(I2{}, No{});

    // ================================================================================================
    // 3x3
    // ================================================================================================
    RB3_STAMP(ts1);
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    int bpp[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) bpp[dy] = ((y0 + wave + dy) % PH) * PW + l31;
    auto px_off = [&](int tap, int half) {      // 16-B chunk 2 h + half of the lane's pixel: channels 8 h + 4 half .. + 3 of the slab
      const int p = bpp[tap / 3] + (tap % 3);
      return p * 16 + ((((2 * h + half) ^ ((p >> 2) & 3))) << 2);
    };
    S3 xc3 = split8(*reinterpret_cast<const f32x4*>(patch + px_off(0, 0)), *reinterpret_cast<const f32x4*>(patch + px_off(0, 1)));
#pragma unroll 1
    for (int cc = 0; cc < SL; ++cc) {
      const float* pslab = patch + cc * (PP * 16);
      static_for<0, 9>([&](auto T) {
        constexpr int t = decltype(T)::value;
        dma(U0 + cc * 9 + t + 2, (t + 2) % RING);
        read_rest(Wc, t % RING);
        read_w1(Wn0, (t + 1) % RING, 0);
        f32x4 N0, N1;                            // the next tap's pixel fragment, under this tap's MFMAs
        if constexpr (t < 8) {
          N0 = *reinterpret_cast<const f32x4*>(pslab + px_off(t + 1, 0));
          N1 = *reinterpret_cast<const f32x4*>(pslab + px_off(t + 1, 1));
        } else {
          const float* nslab = patch + min(cc + 1, SL - 1) * (PP * 16);
          N0 = *reinterpret_cast<const f32x4*>(nslab + px_off(0, 0));
          N1 = *reinterpret_cast<const f32x4*>(nslab + px_off(0, 1));
        }
        static_assert(NT == 3, "mac6x3");
        mac6x3(Wc.w[0], Wc.w[1], Wc.w[2], xc3, acc[0], acc[1], acc[2]);
        const S3 xn3 = split8(N0, N1);           // the next tap's split, between this tap's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NT + 2, 0);
#pragma unroll
        for (int q = 0; q < 6 * NT - 1; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) Wc.w[0][p] = Wn0[p];
        xc3 = xn3;
        sync(VM0{});
      });
    }

    // ================================================================================================
    // tail: the accumulators, converted once, are the B operand: k-step s of channel tile jt = registers 8 s .. 8 s + 7
    // ================================================================================================
    RB3_STAMP(ts2);
    S3 Bop[2 * NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 lo4, hi4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + CH + 32 * jt + 8 * (2 * s + q) + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) (q ? hi4 : lo4)[e] = fmaxf(acc[jt][8 * s + 4 * q + e] + bv[e], 0.0f);
        }
        Bop[2 * jt + s] = split8(lo4, hi4);
      }
    // output / residual addressing AFTER the quad transpose (rb_common.h): access j of an output tile is pixel 4 (l31 / 4) + j of
    // this wave's tile row, 16-B chunk 2 (l31 % 4) + h of its 32 channels
    const int oy = y0 + wave, ox4 = x0 + (l31 & ~3), li = l31 & 3;
    unsigned poff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      poff[j] = (oy < a.H && ox4 + j < a.W) ? ((unsigned)((n * a.H + oy) * a.W + ox4 + j) * (unsigned)(C * 4) + (unsigned)(2 * li + h) * 16u) : kOOB;
    static_for<0, C / 32>([&](auto OT) {
      constexpr int ot = decltype(OT)::value;
      f32x4 R[4];
      f32x16 acc2;
      static_for<0, 2>([&](auto HF) {
        constexpr int half = decltype(HF)::value;
        constexpr int u = U0 + U1 + 2 * ot + half;
        dma((u + 2) % UT, (u + 2) % RING);
        if constexpr (half == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) R[q] = buf_load(xs, poff[q], ot * 128);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
        }
        read_rest(Wc, u % RING);
        read_w1(Wn0, (u + 1) % RING, 0);
        // the unit's three row tiles are the k-steps NT * half + (0 .. NT - 1) of this 32-channel output tile
        static_for<0, NT>([&](auto S) {
          constexpr int s = decltype(S)::value;
          acc2 = mac6(Wc.w[s], Bop[NT * half + s], acc2);
        });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) Wc.w[0][p] = Wn0[p];
        if constexpr (half == 1) {
          f32x4 T[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + 2 * CH + 32 * ot + 8 * q + 4 * h);
            T[q] = f32x4{acc2[4 * q], acc2[4 * q + 1], acc2[4 * q + 2], acc2[4 * q + 3]} + bv;
          }
          quad_transpose(T, li);
#pragma unroll
          for (int j = 0; j < 4; ++j) buf_store(ys, T[j] + R[j], poff[j], ot * 128);
          if constexpr (u == UT - 1) {
            // the next tile's first two K stages are issued LAST, so that this step's wait can leave them (and the stores) in
            // flight: 4 stores + 4 loads per fragment tile
            __builtin_amdgcn_sched_barrier(0);
            if (tile + 1 < t_hi) {
              head_setup(tile + 1);
              if (two) sync(std::integral_constant<int, 12>{});
              else sync(std::integral_constant<int, 8>{});
            } else {
              sync(VM4{});
            }
          } else {
            sync(VM4{});
          }
        } else {
          sync(VM4{});                           // the four residual loads of this output tile stay in flight: they are used a step later
        }
      });
    });
#ifdef SNTC_DIAG
    RB3_STAMP(ts3);
    tph[0] += ts1 - ts0; tph[1] += ts2 - ts1; tph[2] += ts3 - ts2; tph[3] += 1;
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SNTC_DIAG
  __syncthreads();
  if (tid == 0)
    for (int k = 0; k < 4; ++k) a.y[blockIdx.x * 4 + k] = (float)tph[k];
#endif
}

// wpack3[u][plane][row][32 B]: the pre-split units in LDS image order, from the Keras kernels (rb_fused.hip's rb_pack_kernel
// in the k order of a 32x32x16 operand; the tail in the order the accumulator registers present the hidden channels)
template <int C>
__global__ void __launch_bounds__(256) rb3_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                        const float* __restrict__ w2, __bf16* __restrict__ wpack) {
  using K = RBCfg<C>;
  constexpr int CH = K::CH, NT = K::NT;
  const int per_plane = CH * 16;                 // bf16 elements of one plane of one unit
  const int total = K::UT * per_plane;           // one thread per (unit, row, position): it writes the three planes
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int u = idx / per_plane;
    const int rem = idx - u * per_plane;
    const int row = rem >> 4, pos = rem & 15;
    const int hh = (pos >> 3) ^ ((row >> 3) & 1);          // which k half sits in this 16-B half of the row
    const int j = pos & 7;
    float v;
    if (u < K::U0) {
      v = w0[(size_t)(16 * u + 8 * hh + j) * CH + row];
    } else if (u < K::U0 + K::U1) {
      const int s = u - K::U0, cc = s / 9, t = s - cc * 9;
      v = w1[((size_t)t * CH + 16 * cc + 8 * hh + j) * CH + row];
    } else {
      const int s = u - K::U0 - K::U1, ot = s >> 1, half = s & 1;
      const int st = NT * half + (row >> 5);               // k-step 0 .. 5 of K = c/2
      const int ch = 32 * (st >> 1) + 16 * (st & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
      v = w2[(size_t)ch * C + 32 * ot + (row & 31)];
    }
    __bf16 x0, x1, x2;
    split1(v, &x0, &x1, &x2);
    __bf16* dst = wpack + (size_t)u * 3 * per_plane + rem;
    dst[0] = x0;
    dst[per_plane] = x1;
    dst[2 * per_plane] = x2;
  }
}

constexpr size_t kLds3 = (size_t)RBCfg<192>::PATCH * 4 + (size_t)kRing3 * kUnit3 + (size_t)RBCfg<192>::BIAS * 4;
static_assert(kLds3 <= 163840, "patch + ring + biases must fit the CU's LDS");

constexpr int kMaxDev3 = 16;
std::once_flag g_once3[kMaxDev3];
int g_rc3[kMaxDev3];

}  // namespace

int rb3_init() {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev3) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the ResidualBlock tables");
  std::call_once(g_once3[dev], [&] {
    g_rc3[dev] = [&]() -> int {
      SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rb3_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds3));
      return SNTC_OK;
    }();
  });
  return g_rc3[dev];
}

size_t rb3_pack_bytes() { return (size_t)RBCfg<192>::UT * kUnit3; }

int rb3_pack(const float* w0, const float* w1, const float* w2, void* wpack3, hipStream_t s) {
  using K = RBCfg<192>;
  hipLaunchKernelGGL(rb3_pack_kernel<192>, dim3((K::UT * K::CH * 16 + 255) / 256), dim3(256), 0, s, w0, w1, w2,
                     reinterpret_cast<__bf16*>(wpack3));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ResidualBlock split-precision weight packing");
  return SNTC_OK;
}

int rb3_launch(const RBArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL(rb3_kernel<192>, dim3(grid), dim3(512), kLds3, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ResidualBlock (bf16 x 3) launch");
  return SNTC_OK;
}

}  // namespace sntc

// bf3_gemm.hip -- the split-precision ("bf16 x 3") gather GEMM on PRE-SPLIT operands: every fp32 value of the activations and
// of the weights is kept as three bfloat16 terms hi + mid + lo (24 mantissa bits together), and a product is accumulated in
// fp32 from its six significant cross terms on v_mfma_f32_32x32x16_bf16 (6 MFMAs of 32 cycles for K = 16 against 8 fp32 MFMAs
// of 64 cycles: 2.67 x the fp32-MFMA rate; measured loop ceiling on this structure 1.76 x the fp32 PEAK,
// tools/microbench/gemm_ceiling.hip).  Same contraction as csrc/gather_gemm.hip,
//     out[m, col] = sum_t sum_c x[src(m, t), c] * Wp[col][t * Cin + c],
// same phase-grouped transposed convolutions, same stream-K schedule with chain continuation -- a different inner loop:
//
//  * operands arrive split (format "S3": per pixel and 16-channel slab 96 B = [hi x 16 | mid x 16 | lo x 16] bf16; the weights
//    are packed the same way per (column, 16-deep K stage)), so the loop contains NO conversion: a stage goes from L2 / HBM
//    straight into LDS with buffer_load ... lds (16 B per lane, no staging registers, no ds_write);
//  * workgroup = 8 waves as 4 x 2, wave tile 64 x 128 (256 x 256 per workgroup) or 64 x 64 (256 x 128): at 6 B per element a
//    128 x 128 tile would need 14 TB/s of L2 -> LDS traffic (the round-2 experiment's bound), 256 x 256 needs half of that;
//  * LDS image of a stage: row-major [row][96 B]; the two 16-B halves of a plane are swapped on rows 8..15 (mod 16), applied on
//    the SOURCE side (the DMA destination is linear), which makes the fragment ds_read_b128 conflict-free;
//  * three ring slots: in step j the loads of stage j+2 go into the slot stage j-1 left, stage j multiplies, and the step ends
//    once stage j+1 has landed (counted vmcnt) -- one barrier per 16-deep stage.
//
// Not bit-identical to the fp32 path (dropped terms ~2^-24 relative, other summation order inside an MFMA); the parity tests
// hold it to the same 2e-5-vs-float64 bar.  Reference: the arithmetic of tf.nn.conv2d / conv2d_transpose behind
// common/transforms.py:209-232,298-361 (hyper-synthesis, syntheses).
#include <algorithm>
#include <mutex>
#include <type_traits>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr unsigned kOOR = 0x80000000u;          // > any in-range offset: buffers are < 2 GiB (host check)
constexpr int kSpin = 1 << 22;

__device__ __forceinline__ float act_of(float v, int act) {
  switch (act) {
    case SNTC_ACT_RELU: return fmaxf(v, 0.0f);
    case SNTC_ACT_LEAKY_RELU: return v >= 0.0f ? v : 0.2f * v;
    case SNTC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ f32x4 epi_of(f32x4 v, int epi, const f32x4 rs, const float* aux, size_t idx) {
  switch (epi) {
    case SNTC_EPI_ADD: return v + rs;
    case SNTC_EPI_GATE: return rs + *reinterpret_cast<const f32x4*>(aux + idx) * v;
    case SNTC_EPI_MASK_RELU:
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rs[e] > 0.0f ? v[e] : 0.0f;
      return v;
    case SNTC_EPI_MASK_LEAKY:
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rs[e] >= 0.0f ? v[e] : 0.2f * v[e];
      return v;
    default: return v;
  }
}

typedef const GGArgs __attribute__((address_space(4))) KArgs;
__device__ __forceinline__ KArgs& kargs() {
  KArgs* kp = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *kp;
}

struct Piece {
  int gi, mt, nt, k0, k1;
  int consume;   // worker whose published accumulators this piece continues (-1: start from zero)
  int publish;   // 1: the tile is finished by the next worker
};

}  // namespace

// (Tried: weight loads with the non-temporal hint, so that the activation rows the taps of a slab re-read would stay in the 32 KB
// vector L1: 190 -> 178 TFLOP/s-equivalent on the 480 -> 640 layer, dropped.)
template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM* WN * 64, 2) bf3_kernel(const GGArgs a) {
  constexpr int NT = WM * WN * 64;
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int SLOT = (BM + BN) * 96;                    // bytes per ring slot
  constexpr int A_CH = BM * 6 / NT;                       // 16-B chunks of A per thread and stage
  constexpr int B_CH = (BN * 6 + NT - 1) / NT;            // of B (the last round may cover only the first waves)
  static_assert(BM * 6 % NT == 0, "A chunks must divide evenly over the threads");
  constexpr int EPW = 32 * 32;                            // floats of epilogue staging per wave (one 32 x 32 accumulator tile)
  static_assert(SLOT >= NW * EPW * 4, "epilogue staging must fit in ring slot 2 (slots 0 and 1 take the next piece's first stages)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  int4* rinfo_all = reinterpret_cast<int4*>(smem + 3 * SLOT);     // [2][BM] (n, qy, qx, valid) of the current / next tile
  typedef __attribute__((address_space(3))) void lds_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  // ---------------------------------------------------------------- the worker's pieces (stream-K: head, whole tiles, tail)
  int sk_head_t = -1, sk_head_k1 = 0, sk_tail_t = -1, sk_tail_k0 = 0, sk_cur = 0, sk_last = -1, sk_phase = 0, wl = 0;
  if (a.sk) {
    const int w = blockIdx.x;
    wl = (w & 7) * (a.nworkers >> 3) + (w >> 3);
    const int u_lo = (int)(a.units * wl / a.nworkers), u_hi = (int)(a.units * (wl + 1) / a.nworkers);
    // Unit order: COLUMN tile outermost, row strips inside (u = ntm * unit0(g) + nt * ntm * steps + mt * steps + k): the
    // workers of one XCD own a contiguous eighth of the range, i.e. one or two column tiles, whose weight rows (a 128-column
    // tile of the 480 -> 640 layer is 3.3 MB in S3) then stay in that XCD's 4 MB L2 while the strips stream past.
    // a.order == 1: the fp32 kernel's strip-major order (column tile fastest), kept for the A/B.
    auto locate = [&](int u, int* t, int* k, int* steps) {
      int gi = 0;
      if (a.order == 1) {
        const int mt = u / a.ups;
        const int r = u - mt * a.ups;
#pragma unroll
        for (int i = 1; i < kMaxGroups; ++i)
          if (i < a.ngroups && r >= (int)a.g[i].unit0) gi = i;
        const int r2 = r - (int)a.g[gi].unit0;
        const int nt = r2 / a.g[gi].steps;
        *t = mt * a.tps + a.g[gi].tile0 + nt;
        *k = r2 - nt * a.g[gi].steps;
      } else {
#pragma unroll
        for (int i = 1; i < kMaxGroups; ++i)
          if (i < a.ngroups && u >= (int)a.g[i].unit0 * a.ntm) gi = i;
        const int r = u - (int)a.g[gi].unit0 * a.ntm;
        const int per = a.ntm * a.g[gi].steps;
        const int nt = r / per;
        const int r2 = r - nt * per;
        const int mt = r2 / a.g[gi].steps;
        *t = (a.g[gi].tile0 + nt) * a.ntm + mt;
        *k = r2 - mt * a.g[gi].steps;
      }
      *steps = a.g[gi].steps;
    };
    if (u_hi > u_lo) {
      int tF, kF, sF, tL, kL, sL;
      locate(u_lo, &tF, &kF, &sF);
      locate(u_hi - 1, &tL, &kL, &sL);
      sk_cur = tF;
      sk_last = tL;
      if (kF > 0) { sk_tail_t = tF; sk_tail_k0 = kF; sk_cur = tF + 1; }
      if (kL + 1 < sL) { sk_head_t = tL; sk_head_k1 = kL + 1; sk_last = tL - 1; }
    } else {
      sk_phase = 3;
    }
  }
  auto tile_of = [&](int t, Piece* p) {
    KArgs& a = kargs();
    const int mt = a.order == 1 ? t / a.tps : t % a.ntm;
    const int r = a.order == 1 ? t - mt * a.tps : t / a.ntm;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
      if (i < a.ngroups && r >= a.g[i].tile0) gi = i;
    p->gi = gi; p->mt = mt; p->nt = r - a.g[gi].tile0;
  };
  // Order of a worker's pieces: head (published, depends on nothing), whole tiles, tail (continues the previous worker's
  // chain: its head was computed first thing, so the tail never waits).  Whole-tiles-first was tried for lockstep L2 sharing:
  // the tails then wait for heads of uneven length and the remainder phase doubles (HS2 172 -> 124 TFLOP/s-equivalent).
  auto next_piece = [&](Piece* p) -> bool {
    KArgs& a = kargs();
    if (!a.sk) return false;
    if (sk_phase == 0) {
      sk_phase = 1;
      if (sk_head_t >= 0) {
        tile_of(sk_head_t, p);
        p->k0 = 0; p->k1 = sk_head_k1; p->consume = -1; p->publish = 1;
        return true;
      }
    }
    if (sk_phase == 1) {
      if (sk_cur <= sk_last) {
        tile_of(sk_cur++, p);
        p->k0 = 0; p->k1 = a.g[p->gi].steps; p->consume = -1; p->publish = 0;
        return true;
      }
      sk_phase = 2;
    }
    if (sk_phase == 2) {
      sk_phase = 3;
      if (sk_tail_t >= 0) {
        tile_of(sk_tail_t, p);
        p->k0 = sk_tail_k0; p->k1 = a.g[p->gi].steps; p->consume = wl - 1; p->publish = 0;
        return true;
      }
    }
    return false;
  };

  Piece P;
  bool have;
  if (a.sk) {
    have = next_piece(&P);
  } else {                                         // static: one workgroup per tile, XCD-aware order (column tile fastest)
    int gi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
      if (i < a.ngroups && (int)blockIdx.x >= a.g[i].blk0) gi = i;
    const int lb = blockIdx.x - a.g[gi].blk0;
    const int ntn = a.g[gi].ntn;
    const int full = a.ntm & ~7;
    if (lb < full * ntn) {
      const int l = lb >> 3;
      P.mt = (l / ntn) * 8 + (lb & 7);
      P.nt = l % ntn;
    } else {
      const int r = lb - full * ntn, rem = a.ntm - full;
      P.mt = full + r % rem;
      P.nt = r / rem;
    }
    P.gi = gi; P.k0 = 0; P.k1 = a.g[gi].steps; P.consume = -1; P.publish = 0;
    have = true;
  }
  if (!have) return;

  // ---------------------------------------------------------------- loader state
  const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ws = xs;
  // chunk q = i * NT + tid of a stage lands at LDS byte 16 q (linear DMA destination): row q / 6, part q % 6 = (plane, half
  // position); the lane fetches the source half that belongs there: position ^ ((row >> 3) & 1)
  int a_row[A_CH];
  unsigned a_part[A_CH];              // byte offset of the lane's chunk inside the 96-B block of a (pixel, slab)
  int a_iy0[A_CH], a_ix0[A_CH];
  unsigned a_img[A_CH], a_off[A_CH];
  unsigned b_off[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int q = i * NT + tid, row = q / 6, part = q - row * 6;
    a_row[i] = row;
    a_part[i] = (unsigned)((part >> 1) * 32 + (((part & 1) ^ ((row >> 3) & 1)) << 4));
  }
  int ld_stage = 0, ld_t = 0, ld_cc = 0, ld_ty = 0, ld_tx = 0, g_T = 1, g_tw = 1;
  int n0 = 0;
  const int my_b = [&]() {                       // B instructions this wave issues per stage
    int nb = 0;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) nb += (i * NT + wave * 64 < BN * 6) ? 1 : 0;
    return nb;
  }();

  auto set_tap = [&](int ty, int tx) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int iy = a_iy0[i] + ty * a.tstep;
      const int ix = a_ix0[i] + tx * a.tstep;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned pix = ((unsigned)(iy * a.W + ix) * (unsigned)a.Cin * 6u) & 0x7fffffffu;
      a_off[i] = (a_img[i] + pix) | (ok ? 0u : kOOR);
    }
  };
  auto write_rinfo = [&](const Piece& p, int rb) {
    KArgs& a = kargs();
    int4* rinfo = rinfo_all + rb * BM;
    const int mbase = p.mt * BM;
    const int q0y = a.g[p.gi].q0y, q0x = a.g[p.gi].q0x;
    for (int r = tid; r < BM; r += NT) {
      const int m = mbase + r;
      int4 ri = make_int4(0, 0, 0, 0);
      if (m < a.M) {
        const int per = a.Qh * a.Qw;
        const int n = m / per;
        const int rem = m - n * per;
        const int qy = rem / a.Qw;
        ri = make_int4(n, qy + q0y, rem - qy * a.Qw + q0x, 1);
      }
      rinfo[r] = ri;
    }
  };
  auto init_loader = [&](const Piece& p, int rb) {
    KArgs& a = kargs();
    const int4* rinfo = rinfo_all + rb * BM;
    const auto& G = a.g[p.gi];
    g_T = G.T;
    g_tw = G.tw;
    ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.wp), 0, G.Ncol * G.K * 6, 0x00020000);
    n0 = p.nt * BN;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int4 ri = rinfo[a_row[i]];
      a_iy0[i] = ri.y * a.sA + a.offy;
      a_ix0[i] = ri.z * a.sA + a.offx;
      a_img[i] = ri.w ? (unsigned)ri.x * (unsigned)(a.H * a.W) * (unsigned)a.Cin * 6u + a_part[i] : kOOR;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {   // rows past Ncol re-read the last column (finite, discarded)
      const int q = i * NT + tid, row = q / 6, part = q - row * 6;
      const int brow = min(n0 + row, G.Ncol - 1);
      b_off[i] = (unsigned)brow * (unsigned)(G.K / kStage) * 96u + (unsigned)((part >> 1) * 32 + (((part & 1) ^ ((row >> 3) & 1)) << 4));
    }
    ld_stage = p.k0;
    ld_cc = p.k0 / g_T;
    ld_t = p.k0 - ld_cc * g_T;
    ld_ty = ld_t / g_tw;
    ld_tx = ld_t - ld_ty * g_tw;
    set_tap(ld_ty, ld_tx);
  };
  auto issue = [&](int slot) {                  // this wave's share of one stage, L2 / HBM -> LDS
    char* base = ring + slot * SLOT + wave * 1024;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(ld_cc) * 96u;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)(base + i * NT * 16), 16, (int)a_off[i], (int)soff, 0, 0);
    const unsigned wsoff = (unsigned)__builtin_amdgcn_readfirstlane(ld_stage) * 96u;
    char* bb = base + BM * 96;
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      if ((BN * 6) % NT == 0 || i * NT + wave * 64 < BN * 6)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)(bb + i * NT * 16), 16, (int)b_off[i], (int)wsoff, 0, 0);
    // next stage: channel slab outermost, taps inside (k = cc * T * 16 + t * 16 + c), branch-free
    ++ld_stage;
    const int row_end = (ld_tx + 1 == g_tw) ? 1 : 0;
    const int tap_end = (ld_t + 1 == g_T) ? 1 : 0;
    ld_tx = row_end ? 0 : ld_tx + 1;
    ld_ty = tap_end ? 0 : ld_ty + row_end;
    ld_t = tap_end ? 0 : ld_t + 1;
    ld_cc += tap_end;
    set_tap(ld_ty, ld_tx);
  };
  // leave at most `stages` of this wave's stages in flight
  auto wait_stages = [&](auto S) {
    constexpr int s = decltype(S)::value;
    if (s == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (my_b == B_CH) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(s * (A_CH + B_CH)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(s * (A_CH + B_CH - 1)) : "memory");
    }
  };

  // ---------------------------------------------------------------- fragments + MFMA
  const int hoff = (h ^ ((l31 >> 3) & 1)) << 4;
  const int fa = (wm * TM * 32 + l31) * 96 + hoff;
  const int fb = (BM + wn * TN * 32 + l31) * 96 + hoff;
  struct Frag {
    bf16x8 a[3][TM];
    bf16x8 b[3][TN];
  };
  auto read_frag = [&](Frag& F, int slot) {
    const char* base = ring + slot * SLOT;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) F.a[p][i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 32 * 96 + p * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) F.b[p][j] = *reinterpret_cast<const bf16x8*>(base + fb + j * 32 * 96 + p * 32);
    }
  };
  f32x16 acc[TM][TN];
  auto mfma6 = [&](const Frag& F) {           // smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[PA[t]][i], F.b[PB[t]][j], acc[i][j], 0, 0, 0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---------------------------------------------------------------- the piece loop
  int rb = 0;
  write_rinfo(P, rb);
  __syncthreads();
  init_loader(P, rb);
  if (P.k1 - P.k0 > 0) issue(0);
  if (P.k1 - P.k0 > 1) issue(1);
  while (true) {
    const int n = P.k1 - P.k0;
    if (P.consume >= 0) {
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(a.sk_flags + P.consume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > kSpin) {           // never hang, never trap: flag the launch (sntc_conv_status) and carry on
            __hip_atomic_fetch_or(kargs().status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          a.sk_slab + (size_t)P.consume * (TM * TN * 16 * NT), 0, TM * TN * 16 * NT * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(sr, tid * 16, ((i * TN + j) * 4 + q) * NT * 16, 0);
            const f32x4 f = __builtin_bit_cast(f32x4, v);
            acc[i][j][4 * q] = f[0]; acc[i][j][4 * q + 1] = f[1]; acc[i][j][4 * q + 2] = f[2]; acc[i][j][4 * q + 3] = f[3];
          }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }

    // stage 0 must have landed before the first step; stage 1 may still fly.  (Everything older than the two stage issues --
    // the previous piece's epilogue stores were issued AFTER them -- only makes this wait longer, never shorter.)
    if (n > 1) wait_stages(I1{});
    else wait_stages(I0{});
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    Frag F;
    int s_cur = 0, s_n1 = 1, s_n2 = 2;
    auto step = [&](auto LD, auto MORE) {
      if (decltype(LD)::value) issue(s_n2);              // stage j+2 into the slot stage j-1 left at the last barrier
      read_frag(F, s_cur);
      mfma6(F);
      __builtin_amdgcn_sched_barrier(0);
      // stage j+1 has landed; stage j+2 (issued above) may still fly
      if (decltype(LD)::value) wait_stages(I1{});
      else wait_stages(I0{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (decltype(MORE)::value) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const int t = s_cur; s_cur = s_n1; s_n1 = s_n2; s_n2 = t;
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int j = 0;
    for (; j + 2 < n; ++j) step(Yes{}, Yes{});
    if (n - j == 2) { step(No{}, Yes{}); ++j; }
    if (n - j == 1) { step(No{}, No{}); ++j; }

    // ---- the next piece's first two stages go in flight (ring slots 0 and 1) before this piece's results are stored
    Piece Q;
    const bool more = next_piece(&Q);
    const int n0d = n0;
    const int4* rinfo = rinfo_all + rb * BM;
    __syncthreads();                  // every wave has read its last fragments: the whole ring is free
    if (more) {
      write_rinfo(Q, rb ^ 1);
      __syncthreads();
      init_loader(Q, rb ^ 1);
      if (Q.k1 - Q.k0 > 0) issue(0);
      if (Q.k1 - Q.k0 > 1) issue(1);
    }

    // ---- finish the piece that just ran
    KArgs& a = kargs();
    const auto& Gd = a.g[P.gi];
    if (P.publish) {
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          a.sk_slab + (size_t)wl * (TM * TN * 16 * NT), 0, TM * TN * 16 * NT * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j2 = 0; j2 < TN; ++j2)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j2][4 * q], acc[i][j2][4 * q + 1], acc[i][j2][4 * q + 2], acc[i][j2][4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sr, tid * 16, ((i * TN + j2) * 4 + q) * NT * 16, 0);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.sk_flags + wl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // each wave transposes one 32 x 32 accumulator tile at a time through a private 4 KB slice of ring slot 2, so that a
      // lane owns 4 consecutive channels of one pixel (16-B bias / residual reads and stores)
      float* stage = reinterpret_cast<float*>(ring + 2 * SLOT) + wave * EPW;
      const int c4 = (lane & 7) << 2;
      const int rsub = lane >> 3;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j2 = 0; j2 < TN; ++j2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[i][j2][r];
          const int col = n0d + (wn * TN + j2) * 32 + c4;
          const bool col_ok = col < Gd.Ncol;
          unsigned ce = 0;
          f32x4 bv = {0.f, 0.f, 0.f, 0.f};
          if (col_ok) {
            ce = Gd.cols[col];
            if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + (ce & 0xffff));
          }
          const int ch = ce & 0xffff;
          const int oyo = (int)((ce >> 24) & 0xff) - 128;
          const int oxo = (int)((ce >> 16) & 0xff) - 128;
          for (int rp = 0; rp < 32; rp += 8) {
            const int rloc = rp + rsub;
            const int4 ri = rinfo[(wm * TM + i) * 32 + rloc];
            const int oy = ri.y * a.sO + oyo;
            const int ox = ri.z * a.sO + oxo;
            if (!col_ok || !ri.w || (unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
            const size_t idx = (((size_t)ri.x * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + rloc * 32 + c4) + bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_of(v[e], a.act);
            if (a.epi != SNTC_EPI_STORE) v = epi_of(v, a.epi, *reinterpret_cast<const f32x4*>(a.res + idx), a.aux, idx);
            *reinterpret_cast<f32x4*>(a.y + idx) = v;
          }
        }
    }
    if (!more) break;
    P = Q;
    rb ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 NHWC -> S3: per pixel and 16-channel slab [hi x 16 | mid x 16 | lo x 16] bfloat16 (96 B).  One thread per 4 channels.
// Optional fused dequantisation: v = symbols + mu (mu = first half of the hyper-synthesis output, mshyper/models.py:278-279).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) split3_kernel(const float* __restrict__ x, const int* __restrict__ symbols,
                                                     const float* __restrict__ hyper, int64_t npix, int c,
                                                     __bf16* __restrict__ out, float* __restrict__ y_hat) {
  const int64_t total = npix * (c >> 2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / (c >> 2);
    const int c4 = (int)(i - p * (c >> 2)) << 2;
    f32x4 v;
    if (symbols) {
      const int4 s = *reinterpret_cast<const int4*>(symbols + p * c + c4);
      const f32x4 mu = *reinterpret_cast<const f32x4*>(hyper + p * 2 * c + c4);
      v = f32x4{(float)s.x + mu[0], (float)s.y + mu[1], (float)s.z + mu[2], (float)s.w + mu[3]};
      if (y_hat) *reinterpret_cast<f32x4*>(y_hat + p * c + c4) = v;
    } else {
      v = *reinterpret_cast<const f32x4*>(x + p * c + c4);
    }
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hh = (__bf16)v[e];
      const float r1 = v[e] - (float)hh;
      const __bf16 mm = (__bf16)r1;
      hi[e] = hh; mid[e] = mm; lo[e] = (__bf16)(r1 - (float)mm);
    }
    __bf16* o = out + (p * (c >> 4) + (c4 >> 4)) * 48 + (c4 & 15);
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + 16) = mid;
    *reinterpret_cast<bf16x4*>(o + 32) = lo;
  }
}

// ---------------------------------------------------------------------------------------------
// variants + launch (ids continue csrc/gather_gemm.hip's: 11 = 256 x 256, 12 = 256 x 128; both 512 threads, one workgroup per CU)
// ---------------------------------------------------------------------------------------------
static const void* bf3p_kernel(int v) {
  switch (v) {
    case 11: return reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 4>);
    case 12: return reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 2>);
    default: return nullptr;
  }
}

int bf3p_variant_bm(int v) { return 256; }
int bf3p_variant_bn(int v) { return v == 11 ? 256 : 128; }
size_t bf3p_sk_slab_floats(int v) { return (size_t)(v == 11 ? 8 : 4) * 16 * 512; }

static size_t bf3p_lds_bytes(int v) { return (size_t)3 * (bf3p_variant_bm(v) + bf3p_variant_bn(v)) * 96 + 2 * bf3p_variant_bm(v) * sizeof(int4); }

static std::once_flag g_bf3p_once[16];
static int g_bf3p_rc[16];

int bf3p_init() {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the residency tables");
  std::call_once(g_bf3p_once[dev], [&] {
    g_bf3p_rc[dev] = SNTC_OK;
    for (int v : {11, 12}) {
      hipError_t e = hipFuncSetAttribute(bf3p_kernel(v), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf3p_lds_bytes(v));
      if (e != hipSuccess) g_bf3p_rc[dev] = hip_fail(e, "bf3p_init");
    }
  });
  return g_bf3p_rc[dev];
}

int bf3p_launch(int variant, const GGArgs& args, int nblocks, hipStream_t stream) {
  const void* fn = bf3p_kernel(variant);
  if (!fn) return fail(SNTC_ERR_UNSUPPORTED, "unknown pre-split bf16 x 3 tile variant");
  int rc = bf3p_init();
  if (rc) return rc;
  GGArgs a = args;
  void* params[] = {&a};
  hipError_t e = hipLaunchKernel(fn, dim3(nblocks), dim3(512), params, bf3p_lds_bytes(variant), stream);
  if (e != hipSuccess) return hip_fail(e, "pre-split bf16 x 3 gather-GEMM launch");
  return SNTC_OK;
}

}  // namespace sntc

using namespace sntc;

extern "C" int sntc_split3(const float* x, int64_t npix, int c, void* out, void* stream) {
  if (!x || !out) return fail(SNTC_ERR_BAD_SHAPE, "sntc_split3: null argument");
  if (npix < 1 || c < 16 || c % 16) return fail(SNTC_ERR_BAD_SHAPE, "sntc_split3: channels must be a positive multiple of 16");
  const int64_t total = npix * (c >> 2);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (const int*)nullptr, (const float*)nullptr, npix, c,
                     reinterpret_cast<__bf16*>(out), (float*)nullptr);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_dequant_split3(const int32_t* symbols, const float* hyper, int64_t npix, int c, void* out, float* y_hat,
                                   void* stream) {
  if (!symbols || !hyper || !out) return fail(SNTC_ERR_BAD_SHAPE, "sntc_dequant_split3: null argument");
  if (npix < 1 || c < 16 || c % 16) return fail(SNTC_ERR_BAD_SHAPE, "sntc_dequant_split3: channels must be a positive multiple of 16");
  const int64_t total = npix * (c >> 2);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, reinterpret_cast<const int*>(symbols),
                     hyper, npix, c, reinterpret_cast<__bf16*>(out), y_hat);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

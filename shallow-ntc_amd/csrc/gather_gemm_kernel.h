#pragma once
// gather_gemm_kernel.h -- the one contraction kernel of the codec (the kernel TEMPLATE; instantiated per mode in gg_inst_*.hip,
// launched from gather_gemm.hip): an im2col-free, NHWC gather GEMM on
// the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 64 FLOP/clk/SIMD).
//
// Every Conv2D / Conv2DTranspose / SignalConv2D / GDN norm-pool on the hot path is
//     out[m, col] = sum_t sum_c x[src(m, t), c] * Wp[col][t*Cin + c]          (sntc_internal.h)
// with pixels on the MFMA row (A) side and output columns on the column (B) side.
//
// Workgroup: 256 threads = 4 waves arranged WM x WN; each wave owns TN tiles of 32x32; a K stage is 16 deep.
//   HBM/L2 -> registers: buffer_load_dwordx4 through two wave-uniform descriptors (input, packed weights) with a
//   32-bit per-lane byte offset that changes only when the tap changes and a scalar offset that walks the channel
//   slabs / K stages; rows whose source pixel falls outside the image carry an out-of-range offset and read zeros
//   from the bounds check (no branches, no 64-bit address arithmetic in the loop).  Cin % 16 != 0 (the RGB first
//   layer, reduced-width test nets) takes the dword gather path: one k column per thread, (tap, channel) advanced
//   incrementally, same bounds-check zero fill.
//   registers -> LDS: a ring of THREE 16-deep stages, 64-B rows, 16-B chunks XOR-swizzled by (row >> 2) & 3, so the
//   staging ds_write_b128 and the fragment ds_read_b128 are bank-conflict free without padding.
//   Pipeline, ONE barrier per stage: in step j a wave (1) writes stage j+2 (loaded during step j-1) into the slot that
//   held stage j-1, (2) issues the global loads of stage j+3, (3) multiplies stage j from fragments it prefetched,
//   reading the second half's fragments and then the FIRST fragments of stage j+1 under the MFMAs -- so nothing
//   waits on LDS after the barrier.  Per lane a ds_read_b128 fetches k = 8g+4h..8g+4h+3 (h = lane>>5): MFMA e of
//   k-group g sums k in {8g+e, 8g+4+e}; A and B use the same permutation, so the products pair up.
//   C/D layout: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
//
// Scheduling.  Static mode: one workgroup per (tile, K range); K ranges > 1 leave raw partial sums for
// gg_reduce_kernel (deterministic split-K for layers with few tiles per image).  Stream-K mode: as many workgroups
// as the device keeps resident each take an equal share of the launch's (tile, stage) units.  A worker whose share
// ends inside a tile computes that tile's first stages FIRST and publishes the raw accumulators; the next worker
// finishes the tile LAST, starting its fma chains from those accumulators.  Every output element is therefore the
// same k-ordered chain as in an unsplit tile: results do not depend on the worker count, the batch size or the tile
// shape, and no tile quantisation is left (2160 tiles on 768 slots used to run 3 rounds for 2.81 rounds of work).
// Between tiles the next tile's first two stages are in flight while the current tile's epilogue runs.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOutOfRange = 0x80000000u;   // > any in-range offset: buffers are < 2 GiB (host check)
constexpr int kSpinLimit = 1 << 22;             // bounded wait on a neighbour's hand-off (~seconds), then the launch is flagged invalid

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SNTC_ACT_RELU: return fmaxf(v, 0.0f);
    case SNTC_ACT_LEAKY_RELU: return v >= 0.0f ? v : 0.2f * v;
    case SNTC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// f(integral_constant<int, HI>), f(HI - 1), ..., f(LO): stops after the first call that returns true; says whether one did
template <int HI, int LO, class F>
__device__ __forceinline__ bool first_of_desc(F&& f) {
  if constexpr (HI < LO) {
    return false;
  } else {
    if (f(std::integral_constant<int, HI>{})) return true;
    return first_of_desc<HI - 1, LO>(f);
  }
}

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t rsrc, f32x4 v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)voff, (int)soff, 0);
}

__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, 0, 0));
}

__device__ __forceinline__ f32x4 apply_epilogue(f32x4 v, int epi, const f32x4 rs, const float* aux, size_t idx) {
  switch (epi) {
    case SNTC_EPI_ADD: return v + rs;
    case SNTC_EPI_GATE: return rs + *reinterpret_cast<const f32x4*>(aux + idx) * v;
    case SNTC_EPI_RES_DIV: return rs / v;
    case SNTC_EPI_RES_MUL: return rs * v;
    case SNTC_EPI_RES_DIV_SQRT:
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rs[e] / sqrtf(v[e]);
      return v;
    case SNTC_EPI_RES_MUL_SQRT:
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rs[e] * sqrtf(v[e]);
      return v;
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

__device__ __forceinline__ float apply_epilogue1(float v, int epi, const float* res, const float* aux, size_t idx) {
  switch (epi) {
    case SNTC_EPI_ADD: return v + res[idx];
    case SNTC_EPI_GATE: return res[idx] + aux[idx] * v;
    case SNTC_EPI_RES_DIV: return res[idx] / v;
    case SNTC_EPI_RES_MUL: return res[idx] * v;
    case SNTC_EPI_RES_DIV_SQRT: return res[idx] / sqrtf(v);
    case SNTC_EPI_RES_MUL_SQRT: return res[idx] * sqrtf(v);
    case SNTC_EPI_MASK_RELU: return res[idx] > 0.0f ? v : 0.0f;
    case SNTC_EPI_MASK_LEAKY: return res[idx] >= 0.0f ? v : 0.2f * v;
    default: return v;
  }
}

// The kernel arguments through an OPAQUE pointer to the kernarg segment: loads through it cannot be hoisted out of the
// persistent tile loop, so a phase that runs once per tile (piece bookkeeping, row table, epilogue) re-reads its few
// scalars from the scalar cache instead of pinning ~60 SGPRs (and, once those run out, VGPRs) through the K loop.
typedef const GGArgs __attribute__((address_space(4))) KArgs;
__device__ __forceinline__ KArgs& fresh_args() {
  KArgs* kp = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *kp;
}

// One unit of a workgroup's work: stages [k0, k1) of tile (gi, mt, nt).
struct Piece {
  int gi, mt, nt, k0, k1;
  int consume;   // worker whose published accumulators this piece continues (-1: start from zero)
  int publish;   // 1: the tile is finished by the next worker: leave raw accumulators in sk_slab[self]
  int split;     // static split-K: index of this K range
};

// workgroups per CU the register budget must allow (= waves per SIMD for 256-thread workgroups): what the LDS ring admits
constexpr int gg_waves(int tiles) { return tiles == 1 ? 4 : (tiles <= 3 ? 3 : 2); }

// BF3 (experiment, DESIGN.md 8: "bf16 x 3"): every fp32 operand is split into three bfloat16 terms hi + mid + lo (24
// mantissa bits together) and the product is accumulated in fp32 from the six significant cross terms on
// v_mfma_f32_32x32x16_bf16 -- 6 MFMAs of 32 cycles for K = 16 against 8 fp32 MFMAs of 64 cycles.  The weights are split
// once at pack time, the activations while they are staged into LDS.  Results are NOT bit-identical to the fp32 path
// (the dropped terms are ~2^-24 relative, and the summation order inside an MFMA differs): a fenced decode experiment.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16* hi, __bf16* mid, __bf16* lo) {
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;               // exact
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;              // exact
  *hi = h; *mid = m; *lo = (__bf16)r2;
}

// DMA: the stage tiles go from L2 / HBM straight into the LDS ring (buffer_load ... lds): no staging registers, no
// ds_write, and the data never crosses the vector register file.  The LDS destination of a wave instruction is linear
// (64 lanes x 16 B = 16 rows), so the XOR swizzle is applied to the SOURCE chunk each lane fetches.  Four ring slots: stage
// j+3 is issued in step j, stage j+2 is waited for (counted vmcnt) before the barrier of step j, and its first fragments are
// prefetched in step j+1.
template <int TM, int TN, int WM, int WN, bool VEC, bool PRO, bool BF3 = false, bool DMA = false, int DEEP = 0, bool FUSE2 = false,
          bool COLM = false>
__global__ void __launch_bounds__(256, FUSE2 ? 2 : gg_waves(TM * TN)) gg_kernel(const GGArgs a) {
  static_assert(!BF3 || (VEC && !PRO), "the bf16 x 3 experiment covers the vector path without prologue");
  static_assert(!DMA || (VEC && !PRO && !BF3), "direct-to-LDS staging: vector path, no prologue (nothing can touch the data on the way)");
  static_assert(!FUSE2 || (TM == 1 && TN == 3 && WM == 4 && WN == 1 && VEC && !PRO && !BF3 && !DMA),
                "FUSE2: 3x3 (N = 96) -> 1x1 (96 -> 192) of a ResidualBlock, the 128 x 96 register-staged instance only");
  static_assert(DEEP == 0 || (DMA && (DEEP & (DEEP - 1)) == 0 && DEEP >= 4), "DEEP: ring slots of the direct-to-LDS pipeline, a power of two");
  constexpr int RING = DMA ? (DEEP ? DEEP : 4) : 3;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  // floats per ring slot.  fp32: A rows, then B rows, 16 floats (64 B) each.  BF3: three bf16 planes of A rows, then three of
  // B rows, 16 bf16 (32 B) per row and plane = 96 B per row
  constexpr int SLOT = BF3 ? (BM + BN) * 24 : (BM + BN) * kStage;
  constexpr int A_CH = BM / 64;                       // 16-B chunks per thread per stage (A, vector path)
  constexpr int B_CH = BF3 ? (BN * 6 + 255) / 256 : (BN + 63) / 64;   // BF3: 6 chunks of 16 B per weight row and stage
  constexpr int A_SC = BM / 16;                       // dwords per thread per stage (A, gather path)
  constexpr int EPW = 32 * (TN >= 2 ? 64 : 32);       // floats of epilogue staging per wave
  constexpr bool DBUF = !BF3 && TM * TN <= 8 && TN <= 5;   // fragment double buffering (the two widest 32-row tiles, 96+ accumulator
                                                      // registers and 7-8 fragment quads, run single-buffered)
  constexpr bool RBUF = DBUF && TM * TN <= 3 && !FUSE2;         // second staging register set for a tile's first two stages
  static_assert(RING * SLOT >= 4 * EPW + (DMA ? SLOT : 0), "epilogue staging (and one prefetched stage) must fit in the stage ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ring = reinterpret_cast<float*>(smem);                  // [RING][BM + BN][16]
  int4* rinfo_all = reinterpret_cast<int4*>(ring + RING * SLOT); // [2][BM] (n, qy, qx, valid) of the current / next tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int l31 = lane & 31;
  const int h = lane >> 5;

  // ------------------------------------------------------------------------------------------------------
  // the worker's list of pieces
  // ------------------------------------------------------------------------------------------------------
  // stream-K: [head piece of the LAST tile of the share (published)] [whole tiles] [tail piece of the FIRST tile]
  int sk_head_t = -1, sk_head_k1 = 0;       // global tile id / end stage of the head piece
  int sk_tail_t = -1, sk_tail_k0 = 0;
  int sk_cur = 0, sk_last = -1;             // whole tiles [sk_cur, sk_last]
  int sk_phase = 0;                         // 0 head, 1 whole tiles, 2 tail, 3 done
  int wl = 0;
  if (a.sk) {
    const int w = blockIdx.x;
    wl = (w & 7) * (a.nworkers >> 3) + (w >> 3);      // workers on one XCD (b, b + 8, ...) own one contiguous eighth
    const int u_lo = (int)(a.units * wl / a.nworkers);
    const int u_hi = (int)(a.units * (wl + 1) / a.nworkers);
    // tile order: row strip major, then group, then column tile -- every worker's share mixes the groups (their tiles
    // differ in length, so a group-major order would hand some workers only short, epilogue-heavy tiles) and one strip's
    // input rows serve all groups while they are hot in L2
    // COLM (a compile-time twin of the kernel, single-group plans only: round 3 had this as a run-time branch and the branch
    // moved the whole kernel's register allocation): COLUMN tile outermost, row strips inside (u = nt * ntm * steps + mt * steps
    // + k, tile id = nt * ntm + mt).  The workers of one XCD own a contiguous eighth of the range, i.e. less than one column
    // tile of a five-column layer, whose weight rows then stay in that XCD's 4 MB L2 while the strips stream past -- for
    // layers whose packed weights exceed the L2 (hyper-synthesis 480 -> 640: 11 MB), where the strip-major order streams the
    // whole matrix through every XCD once per strip.
    auto locate = [&](int u, int* t, int* k, int* steps) {
      if constexpr (COLM) {
        const int st = a.g[0].steps;
        const int per = a.ntm * st;
        const int nt = u / per;
        const int r2 = u - nt * per;
        const int mt = r2 / st;
        *t = nt * a.ntm + mt;
        *k = r2 - mt * st;
        *steps = st;
      } else {
        const int mt = u / a.ups;
        const int r = u - mt * a.ups;
        int gi = 0;
#pragma unroll
        for (int i = 1; i < kMaxGroups; ++i)
          if (i < a.ngroups && r >= (int)a.g[i].unit0) gi = i;
        const int r2 = r - (int)a.g[gi].unit0;
        const int nt = r2 / a.g[gi].steps;
        *t = mt * a.tps + a.g[gi].tile0 + nt;
        *k = r2 - nt * a.g[gi].steps;
        *steps = a.g[gi].steps;
      }
    };
    if (u_hi > u_lo) {
      int tF, kF, sF, tL, kL, sL;
      locate(u_lo, &tF, &kF, &sF);
      locate(u_hi - 1, &tL, &kL, &sL);
      sk_cur = tF;
      sk_last = tL;
      if (kF > 0) { sk_tail_t = tF; sk_tail_k0 = kF; sk_cur = tF + 1; }
      if (kL + 1 < sL) { sk_head_t = tL; sk_head_k1 = kL + 1; sk_last = tL - 1; }
      // host guarantee: a share is at least as long as the longest tile, so head and tail are different tiles
    } else {
      sk_phase = 3;
    }
  }

  auto tile_of = [&](int t, Piece* p) {          // global tile id -> (group, row strip, column tile): column fastest
    KArgs& a = fresh_args();
    if constexpr (COLM) {                        // one group; tile id = nt * ntm + mt
      const int nt = t / a.ntm;
      p->gi = 0;
      p->mt = t - nt * a.ntm;
      p->nt = nt;
    } else {
      const int mt = t / a.tps;
      const int r = t - mt * a.tps;
      int gi = 0;
#pragma unroll
      for (int i = 1; i < kMaxGroups; ++i)
        if (i < a.ngroups && r >= a.g[i].tile0) gi = i;
      p->gi = gi;
      p->mt = mt;
      p->nt = r - a.g[gi].tile0;
    }
  };

  auto next_piece = [&](Piece* p) -> bool {
    KArgs& a = fresh_args();
    if (!a.sk) return false;
    if (sk_phase == 0) {
      sk_phase = 1;
      if (sk_head_t >= 0) {
        tile_of(sk_head_t, p);
        p->k0 = 0; p->k1 = sk_head_k1; p->consume = -1; p->publish = 1; p->split = 0;
        return true;
      }
    }
    if (sk_phase == 1) {
      if (sk_cur <= sk_last) {
        tile_of(sk_cur++, p);
        p->k0 = 0; p->k1 = a.g[p->gi].steps; p->consume = -1; p->publish = 0; p->split = 0;
        return true;
      }
      sk_phase = 2;
    }
    if (sk_phase == 2) {
      sk_phase = 3;
      if (sk_tail_t >= 0) {
        tile_of(sk_tail_t, p);
        p->k0 = sk_tail_k0; p->k1 = a.g[p->gi].steps; p->consume = wl - 1; p->publish = 0; p->split = 0;
        return true;
      }
    }
    return false;
  };

  Piece P;
  bool have;
  if (a.sk) {
    have = next_piece(&P);
  } else {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
      if (i < a.ngroups && (int)blockIdx.x >= a.g[i].blk0) gi = i;
    const int lb0 = blockIdx.x - a.g[gi].blk0;
    const int split = lb0 % a.ksplit;          // K range of this block (deterministic split-K)
    const int lb = lb0 / a.ksplit;
    // XCD-aware tile order: blocks b and b + 8 share an XCD (and its 4 MB L2).  Inside one XCD's sequence the column
    // tile runs fastest, so the blocks resident on an XCD cover a few row strips x all column tiles.
    const int ntn = a.g[gi].ntn;
    const int full = a.ntm & ~7;
    if (lb < full * ntn) {
      const int l = lb >> 3;
      P.mt = (l / ntn) * 8 + (lb & 7);
      P.nt = l % ntn;
    } else {                                   // ragged tail: fewer than 8 row strips left
      const int r = lb - full * ntn, rem = a.ntm - full;
      P.mt = full + r % rem;
      P.nt = r / rem;
    }
    const int steps = a.g[gi].steps;
    P.gi = gi;
    P.k0 = (int)(((long long)split * steps) / a.ksplit);
    P.k1 = (int)(((long long)(split + 1) * steps) / a.ksplit);
    P.consume = -1; P.publish = 0; P.split = split;
    have = true;
  }
  if (!have) return;

  // ------------------------------------------------------------------------------------------------------
  // per-piece loader state
  // ------------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t xs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const int r0 = tid >> 2;        // vector path: row 0..63 (+64 i)
  // vector path: 16-B chunk inside the 64-B K slab this lane fetches.  Register staging: lane (tid & 3) fetches chunk
  // tid & 3 and WRITES it to the swizzled LDS position; DMA: the LDS position is the lane's own (linear destination), so
  // the lane fetches the chunk that belongs there
  const int c = DMA ? ((tid & 3) ^ ((r0 >> 2) & 3)) : (tid & 3);
  const int wsw = (c ^ ((r0 >> 2) & 3)) << 2;
  const int kk = tid & 15;        // gather path: k column inside the stage
  const int rs = tid >> 4;        // gather path: row 0..15 (+16 i)
  const int gsw = (((kk >> 2) ^ ((rs >> 2) & 3)) << 2) + (kk & 3);
  constexpr int NROW = VEC ? A_CH : A_SC;

  int a_iy0[NROW], a_ix0[NROW];
  unsigned a_img[NROW];           // byte offset of the row's image (+ chunk), or kOutOfRange for padding rows
  unsigned a_off[NROW];           // vector path: current tap's byte offset per row
  unsigned b_off[B_CH];
  int ld_stage = 0, ld_t = 0, ld_cc = 0;        // next stage to load; vector path: its (tap, channel slab)
  int ld_ty = 0, ld_tx = 0;                     // vector path: the tap's (row, column) in the group's tw-wide tap grid
  int g_t = 0, g_ch = 0, g_ty = 0, g_tx = 0;    // gather path: (tap, channel) of this thread's k column
  int g_T = a.g[P.gi].T, g_tw = a.g[P.gi].tw;   // hot fields of the current piece's group: taps form a dense grid, tap t
                                                // = (t / tw, t % tw), walked incrementally (no table load in the loop)
  __amdgpu_buffer_rsrc_t ws =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[P.gi].wp), 0, a.g[P.gi].Ncol * a.g[P.gi].K * (BF3 ? 6 : 4), 0x00020000);
  int m0 = 0, n0 = 0;

  auto set_tap = [&](int ty, int tx) {
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
      const int iy = a_iy0[i] + ty * a.tstep;
      const int ix = a_ix0[i] + tx * a.tstep;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned pix = ((unsigned)(iy * a.W + ix) * (unsigned)a.Cin * 4u) & 0x7fffffffu;
      // branch-free: in-range sums stay below 2^31 (host check); a padding row (a_img = 2^31) or a tap outside the image
      // gets bit 31 and reads zeros from the descriptor's bounds check
      a_off[i] = (a_img[i] + pix) | (ok ? 0u : kOutOfRange);
    }
  };

  // rinfo of piece `p` into buffer `rb`; caller synchronises before reading it
  auto write_rinfo = [&](const Piece& p, int rb) {
    KArgs& a = fresh_args();
    int4* rinfo = rinfo_all + rb * BM;
    const int mbase = p.mt * BM;
    const int q0y = a.g[p.gi].q0y, q0x = a.g[p.gi].q0x;
    for (int r = tid; r < BM; r += 256) {
      const int m = mbase + r;
      int4 ri = make_int4(0, 0, 0, 0);
      if (m < a.M) {
        const int per = a.Qh * a.Qw;
        const int n = m / per;
        const int rem = m - n * per;
        const int qy = rem / a.Qw;
        ri = make_int4(n, qy + q0y, rem - qy * a.Qw + q0x, 1);   // per-group origin of the macro grid
      }
      rinfo[r] = ri;
    }
  };

  auto init_loader = [&](const Piece& p, int rb) {
    KArgs& a = fresh_args();
    const int4* rinfo = rinfo_all + rb * BM;
    const auto& G = a.g[p.gi];
    g_T = G.T;
    g_tw = G.tw;
    ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.wp), 0, G.Ncol * G.K * (BF3 ? 6 : 4), 0x00020000);
    m0 = p.mt * BM;
    n0 = p.nt * BN;
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
      const int4 ri = rinfo[VEC ? r0 + 64 * i : rs + 16 * i];
      a_iy0[i] = ri.y * a.sA + a.offy;
      a_ix0[i] = ri.z * a.sA + a.offx;
      a_img[i] = ri.w ? (unsigned)ri.x * (unsigned)(a.H * a.W) * (unsigned)a.Cin * 4u + (VEC ? (unsigned)c * 16u : 0u)
                      : kOutOfRange;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {   // rows past Ncol re-read the last column (finite, discarded)
      if (BF3) {                        // chunk q of the tile's stage: weight row q / 6, 16-B part q % 6 = (plane, half)
        const int q = tid + 256 * i;
        const int brow = min(n0 + q / 6, G.Ncol - 1);
        b_off[i] = (unsigned)brow * (unsigned)(G.K / kStage) * 96u + (unsigned)(q % 6) * 16u;
      } else {
        const int brow = min(n0 + r0 + 64 * i, G.Ncol - 1);
        b_off[i] = (unsigned)brow * (unsigned)G.K * 4u + (unsigned)c * 16u;
      }
    }
    ld_stage = p.k0;
    if (VEC) {
      ld_cc = p.k0 / g_T;
      ld_t = p.k0 - ld_cc * g_T;
      ld_ty = ld_t / g_tw;
      ld_tx = ld_t - ld_ty * g_tw;
      set_tap(ld_ty, ld_tx);
    } else {
      const int k = p.k0 * kStage + kk;
      g_t = k / a.Cin;
      g_ch = k - g_t * a.Cin;
      g_ty = g_t / g_tw;
      g_tx = g_t - g_ty * g_tw;
    }
  };

  struct Regs {
    f32x4 a[VEC ? A_CH : 1];
    float s[VEC ? 1 : A_SC];
    f32x4 b[B_CH];
  };

  auto advance_stage = [&]() {
    ++ld_stage;
    if (VEC) {
      // next stage's channel slab / tap, without a branch (the steady-state loop stays one basic block: the scheduler
      // can then place every load, LDS write and fragment read between MFMAs); the per-row offsets are recomputed every
      // stage -- ~10 VALU per row against 1024 MFMA cycles
      // K order of the vector path: channel slab OUTERMOST, taps inside (k = cc * T * 16 + t * 16 + c): the taps of one
      // 16-channel slab re-read the same input pixels (a 5x5 / stride-2 layer touches each ~6 times), so they now do it
      // within T consecutive stages, from L2, instead of T * Cin / 16 stages apart, from HBM
      const int row_end = (ld_tx + 1 == g_tw) ? 1 : 0;
      const int tap_end = (ld_t + 1 == g_T) ? 1 : 0;
      ld_tx = row_end ? 0 : ld_tx + 1;
      ld_ty = tap_end ? 0 : ld_ty + row_end;
      ld_t = tap_end ? 0 : ld_t + 1;
      ld_cc += tap_end;
      set_tap(ld_ty, ld_tx);
    }
  };

  auto load_regs = [&](Regs& R) {
    if (VEC) {
      const unsigned soff = (unsigned)ld_cc * 64u;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) R.a[i] = buf_load(xs, a_off[i], soff);
    } else {
      const bool kok = g_t < g_T;
      const int ty = g_ty * a.tstep, tx = g_tx * a.tstep;
#pragma unroll
      for (int i = 0; i < A_SC; ++i) {
        const int iy = a_iy0[i] + ty;
        const int ix = a_ix0[i] + tx;
        const bool ok = kok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a_img[i] & kOutOfRange);
        const unsigned off = a_img[i] + ((unsigned)(iy * a.W + ix) * (unsigned)a.Cin + (unsigned)g_ch) * 4u;
        R.s[i] = buf_load1(xs, ok ? off : kOutOfRange);
      }
      g_ch += kStage;
      while (g_ch >= a.Cin) {
        g_ch -= a.Cin;
        ++g_t;
        if (++g_tx == g_tw) { g_tx = 0; ++g_ty; }
      }
    }
    const unsigned wsoff = (unsigned)ld_stage * (BF3 ? 96u : 64u);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      if (BF3 ? ((BN * 6) % 256 == 0 || tid + 256 * i < BN * 6) : (BN % 64 == 0 || r0 + 64 * i < BN))
        R.b[i] = buf_load(ws, b_off[i], wsoff);
    advance_stage();
  };

  // DMA: this wave's share of one stage, HBM / L2 -> LDS slot.  Instruction i of wave w lands on rows 16 w + 64 i ... + 15
  // (64 lanes x 16 B, linear); the per-lane source offsets are the register path's (with the source-side swizzle in c)
  const int my_nb = [&]() {                    // B instructions this wave issues per stage (rows 16 w + 64 i < BN)
    int nb = 0;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) nb += ((tid >> 6) * 16 + 64 * i < BN) ? 1 : 0;
    return nb;
  }();
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: the LDS destination (M0) and the
                                                                    // scalar offsets must not make the compiler build waterfall loops
  auto dma_stage = [&](int slot) {
    typedef __attribute__((address_space(3))) void lds_void;
    float* Ab = ring + __builtin_amdgcn_readfirstlane(slot) * SLOT + wave_u * 16 * kStage;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(ld_cc) * 64u;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)(Ab + 64 * i * kStage), 16, (int)a_off[i], (int)soff, 0, 0);
    float* Bb = Ab + BM * kStage;
    const unsigned wsoff = (unsigned)__builtin_amdgcn_readfirstlane(ld_stage) * 64u;
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      if (BN % 64 == 0 || wave_u * 16 + 64 * i < BN)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)(Bb + 64 * i * kStage), 16, (int)b_off[i], (int)wsoff, 0, 0);
    advance_stage();
  };
  // wait until at most K of this wave's DMA stages are outstanding (vmcnt counts instructions, in order); K is a compile-time
  // constant because s_waitcnt takes an immediate
  auto dma_wait = [&](auto K) {
    constexpr int k = decltype(K)::value < 0 ? 0 : decltype(K)::value;
    if (k == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (my_nb == B_CH) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k * (A_CH + B_CH)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k * (A_CH + B_CH - 1)) : "memory");
    }
  };
  auto write_lds = [&](const Regs& R, int slot) {
    float* Ab = ring + slot * SLOT;
    if (BF3) {
      // A: split the four fp32 values of this thread's chunk into three bf16 quads; plane p of row r lives at
      // p * BM * 32 + r * 32 bytes, its two 16-B halves swapped by (r >> 3) & 1 (conflict-free fragment reads)
      char* Ap = reinterpret_cast<char*>(Ab);
      char* Bp = Ap + 3 * BM * 32;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int r = r0 + 64 * i;
        bf16x4 hi, mid, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 x0, x1, x2;
          split3(R.a[i][e], &x0, &x1, &x2);
          hi[e] = x0; mid[e] = x1; lo[e] = x2;
        }
        const int off = r * 32 + (((c >> 1) ^ ((r >> 3) & 1)) << 4) + ((c & 1) << 3);
        *reinterpret_cast<bf16x4*>(Ap + off) = hi;
        *reinterpret_cast<bf16x4*>(Ap + BM * 32 + off) = mid;
        *reinterpret_cast<bf16x4*>(Ap + 2 * BM * 32 + off) = lo;
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        const int q = tid + 256 * i;
        if ((BN * 6) % 256 == 0 || q < BN * 6) {
          const int r = q / 6, part = q % 6, plane = part >> 1, half = part & 1;
          *reinterpret_cast<f32x4*>(Bp + plane * BN * 32 + r * 32 + ((half ^ ((r >> 3) & 1)) << 4)) = R.b[i];
        }
      }
      return;
    }
    float* Bb = Ab + BM * kStage;
    if (VEC) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        f32x4 v = R.a[i];
        if (PRO) {
          if (a.pro == SNTC_PRO_ABS) {
            v[0] = fabsf(v[0]); v[1] = fabsf(v[1]); v[2] = fabsf(v[2]); v[3] = fabsf(v[3]);
          } else if (a.pro == SNTC_PRO_SQUARE) {
            v = v * v;
          }
        }
        *reinterpret_cast<f32x4*>(Ab + (r0 + 64 * i) * kStage + wsw) = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_SC; ++i) {
        float v = R.s[i];
        if (PRO) {
          if (a.pro == SNTC_PRO_ABS) v = fabsf(v);
          else if (a.pro == SNTC_PRO_SQUARE) v = v * v;
        }
        Ab[(rs + 16 * i) * kStage + gsw] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      if (BN % 64 == 0 || r0 + 64 * i < BN)
        *reinterpret_cast<f32x4*>(Bb + (r0 + 64 * i) * kStage + wsw) = R.b[i];
  };

  // fragment addresses: row = 32-row block + l31, chunk (2g + h) ^ ((row >> 2) & 3)
  const int swz = (l31 >> 2) & 3;
  const int fa_row = (wm * TM * 32 + l31) * kStage;
  const int fb_row = (BM + wn * TN * 32 + l31) * kStage;
  const int foff0 = ((0 + h) ^ swz) << 2, foff1 = ((2 + h) ^ swz) << 2;

  struct Frag {
    f32x4 a[TM];
    f32x4 b[TN];
  };
  auto read_frag = [&](Frag& F, int slot, int g) {
    const float* base = ring + slot * SLOT;
    const int off = g ? foff1 : foff0;
#pragma unroll
    for (int i = 0; i < TM; ++i) F.a[i] = *reinterpret_cast<const f32x4*>(base + fa_row + i * 32 * kStage + off);
#pragma unroll
    for (int j = 0; j < TN; ++j) F.b[j] = *reinterpret_cast<const f32x4*>(base + fb_row + j * 32 * kStage + off);
  };

  struct Frag3 {
    bf16x8 a[BF3 ? 3 : 1][TM];
    bf16x8 b[BF3 ? 3 : 1][TN];
  };
  auto read_frag3 = [&](Frag3& F, int slot) {
    const char* base = reinterpret_cast<const char*>(ring + slot * SLOT);
    const int hoff = (h ^ ((l31 >> 3) & 1)) << 4;
#pragma unroll
    for (int p = 0; p < (BF3 ? 3 : 1); ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        F.a[p][i] = *reinterpret_cast<const bf16x8*>(base + p * BM * 32 + (wm * TM * 32 + i * 32 + l31) * 32 + hoff);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        F.b[p][j] = *reinterpret_cast<const bf16x8*>(base + 3 * BM * 32 + p * BN * 32 + (wn * TN * 32 + j * 32 + l31) * 32 + hoff);
    }
  };

  f32x16 acc[TM][TN];
  auto mfma3 = [&](const Frag3& F) {           // smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < (BF3 ? 6 : 0); ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[BF3 ? PA[t] : 0][i], F.b[BF3 ? PB[t] : 0][j], acc[i][j], 0, 0, 0);
  };
  auto mfma_group = [&](const Frag& F) {
    if (SNTC_DBG(a, 64)) return;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = FUSE2 ? __builtin_amdgcn_mfma_f32_32x32x2f32(F.b[j][e], F.a[i][e], acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[i][e], F.b[j][e], acc[i][j], 0, 0, 0);
  };

  // ------------------------------------------------------------------------------------------------------
  // the piece loop
  // ------------------------------------------------------------------------------------------------------
  int rb = 0;                                  // rinfo buffer of the current piece
  Regs R0, R1;
  write_rinfo(P, rb);
  __syncthreads();
  init_loader(P, rb);
  int dma_issued = 0;                          // DMA: stages of the current piece already on their way (slots 0 ...)
  if (DMA) {
    const int n = P.k1 - P.k0;
    for (; dma_issued < RING - 1 && dma_issued < n; ++dma_issued) dma_stage(dma_issued);
  } else {
    const int n = P.k1 - P.k0;
    if (n > 0) load_regs(R0);
    if (RBUF && n > 1) load_regs(R1);
  }

  while (true) {
    const int n = P.k1 - P.k0;
    // ---- accumulators: zero, or the previous worker's published partial sums (stream-K continuation)
    if (P.consume >= 0) {
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(a.sk_flags + P.consume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > kSpinLimit) {
            // a neighbour that never published (it cannot be co-resident: HIP promises neither residency nor dispatch
            // order): never hang and never take the context down -- flag the launch as invalid and carry on with whatever
            // the slab holds; the host reads the sticky word at its next synchronisation point (sntc_conv_status), raises,
            // and can re-run on the static schedule (sntc_conv_set_stream_k(0))
            __hip_atomic_fetch_or(fresh_args().status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      // slab layout [TN * 4 quads][256 threads][4 floats]: 16 B per lane, 1 KB per wave instruction; addressed through a
      // buffer descriptor with constant scalar offsets (no per-store 64-bit address registers)
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          a.sk_slab + (size_t)P.consume * (TM * TN * 16 * 256), 0, TM * TN * 16 * 256 * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = buf_load(sr, (unsigned)tid * 16u, (unsigned)((i * TN + j) * 4 + q) * 4096u);
            acc[i][j][4 * q] = v[0]; acc[i][j][4 * q + 1] = v[1]; acc[i][j][4 * q + 2] = v[2]; acc[i][j][4 * q + 3] = v[3];
          }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }

    Frag F0, F1;
    if (DMA) {
      // ---- direct-to-LDS pipeline: stages 0 .. RING-2 are (being) issued; slot of stage j is j & (RING-1).  Step j issues
      // stage j + RING - 1 into the slot stage j - 1 left at the last barrier and ends once stage j + 2 has landed, so
      // RING - 3 younger stages stay in flight across the barrier (1 with the 4-slot ring; 5 with the 8-slot ring of
      // the DEEP instances, which launches too small to hide the memory latency behind other workgroups are given)
      for (; dma_issued < RING - 1 && dma_issued < n; ++dma_issued) dma_stage(dma_issued);
      // stages 0 and 1 must have landed before the first step (stage 1's fragments are prefetched in step 0)
      using Yes = std::integral_constant<bool, true>;
      using No = std::integral_constant<bool, false>;
      auto wait_first = [&](auto I) {           // issued = min(n, RING - 1) stages; all but the first two may still fly
        constexpr int i = decltype(I)::value;
        if constexpr (i > 2) {
          if (n >= i) { dma_wait(std::integral_constant<int, i - 2>{}); return true; }
        }
        return false;
      };
      if (!first_of_desc<RING - 1, 3>(wait_first)) dma_wait(std::integral_constant<int, 0>{});
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (n > 0) read_frag(F0, 0, 0);
      // REM: steps left including this one when no stage is left to issue (compile-time, for the counted wait); -1 in the
      // steady state
      auto dstep = [&](int j, auto LD, auto PF, auto REM) {
        if (decltype(LD)::value) dma_stage((j + RING - 1) & (RING - 1));   // slot of stage j - 1, free since the last barrier
        read_frag(F1, j & (RING - 1), 1);
        mfma_group(F0);
        if (decltype(PF)::value) read_frag(F0, (j + 1) & (RING - 1), 0);
        mfma_group(F1);
        if (decltype(LD)::value) {
          constexpr int NW = A_CH + B_CH, NR = TM + TN, NM = 4 * TM * TN;
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, NW, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NM - 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // stage j + 2 has landed; the younger ones may still fly
        if constexpr (decltype(LD)::value) dma_wait(std::integral_constant<int, RING - 3>{});
        else dma_wait(std::integral_constant<int, decltype(REM)::value - 3>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      int j = 0;
      for (; j + RING - 1 < n; ++j) dstep(j, Yes{}, Yes{}, std::integral_constant<int, -1>{});
      // the last RING - 1 (or fewer) steps: everything is issued, the counted wait shrinks with the steps left
      auto tail = [&](auto REM) {
        constexpr int rem = decltype(REM)::value;
        if (n - j == rem) {
          if constexpr (rem > 1) dstep(j, No{}, Yes{}, REM);
          else dstep(j, No{}, No{}, REM);
          ++j;
        }
      };
      first_of_desc<RING - 1, 1>([&](auto REM) { tail(REM); return false; });
      dma_issued = 0;
    } else {
    // ---- prologue: stages k0, k0+1 -> ring slots 0, 1; stage k0+2 in flight
    int s_cur = 0, s_n1 = 1, s_n2 = 2;
    if (n > 0) write_lds(R0, 0);
    if (RBUF) {
      if (n > 1) write_lds(R1, 1);
    } else if (n > 1) {
      load_regs(R0);
      write_lds(R0, 1);
    }
    if (n > 2) load_regs(R0);
    __syncthreads();
    if (DBUF && n > 0) read_frag(F0, 0, 0);

    // one step = one stage.  WR: stage j+2 goes from registers into the ring; LD: stage j+3's global loads are issued;
    // PF: the first fragments of stage j+1 are prefetched.  The steady-state steps have all three and no branch.
    auto step = [&](auto WR, auto LD, auto PF) {
      if (decltype(WR)::value && !SNTC_DBG(a, 2)) write_lds(R0, s_n2);        // stage j+2, loaded during step j-1
      if (decltype(LD)::value && !SNTC_DBG(a, 1)) load_regs(R0);              // stage j+3
      if (BF3) {
        Frag3 F3;
        read_frag3(F3, s_cur);
        mfma3(F3);
      } else if (DBUF) {
        if (!SNTC_DBG(a, 8)) read_frag(F1, s_cur, 1);
        mfma_group(F0);
        if (decltype(PF)::value && !SNTC_DBG(a, 8)) read_frag(F0, s_n1, 0);   // under this stage's remaining MFMAs
        mfma_group(F1);
      } else {
        read_frag(F0, s_cur, 0);
        mfma_group(F0);
        read_frag(F0, s_cur, 1);
        mfma_group(F0);
      }
      if (DBUF && VEC && decltype(WR)::value && decltype(LD)::value && decltype(PF)::value) {
        // steady state: order the step's memory instructions BETWEEN its MFMAs (mask 0x8 MFMA, 0x200 DS write, 0x20 VMEM
        // read, 0x100 DS read): an MFMA occupies the pipe for 64 cycles but its issue slot for a few, so everything
        // placed behind the first one is free; the fragment reads get a full MFMA group to land before they are used
        constexpr int NW = A_CH + B_CH, NR = TM + TN, NM = 4 * TM * TN;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, NW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, NW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);                   // every MFMA of the step is issued before the wave waits
      if (!SNTC_DBG(a, 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      const int t = s_cur; s_cur = s_n1; s_n1 = s_n2; s_n2 = t;
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    for (int j = 0; j + 3 < n; ++j) step(Yes{}, Yes{}, Yes{});
    if (n >= 3) step(Yes{}, No{}, Yes{});
    if (n >= 2) step(No{}, No{}, Yes{});
    if (n >= 1) step(No{}, No{}, No{});
    }   // register-staged pipeline

    // ---- the next piece's first stages go in flight before this piece's results are stored
    Piece Q;
    const bool more = next_piece(&Q);
    const int qb = rb ^ 1;
    const int4* rinfo = rinfo_all + rb * BM;
    const int m0d = m0, n0d = n0;
    if (more) {
      write_rinfo(Q, qb);
      __syncthreads();
      init_loader(Q, qb);
      const int nq = Q.k1 - Q.k0;
      if (DMA) {
        // the epilogue stages through the TOP of the ring; the slots below it take the next piece's first stages now
        constexpr int NPF = (RING * SLOT - 4 * EPW) / SLOT < RING - 1 ? (RING * SLOT - 4 * EPW) / SLOT : RING - 1;
        for (dma_issued = 0; dma_issued < NPF && dma_issued < nq; ++dma_issued) dma_stage(dma_issued);
      } else {
        if (nq > 0) load_regs(R0);
        if (RBUF && nq > 1) load_regs(R1);
      }
    }

    // ---- finish the piece that just ran.  Lane constants and kernel arguments of this phase are re-derived from an
    // opaque copy of the thread id / of the kernel-argument pointer: the compiler would otherwise hoist them out of the
    // persistent loop and keep ~40 registers alive through the K loop for values that are used once per tile.
    int te = tid;
    asm volatile("" : "+v"(te));
    const int lane = te & 63, wave = te >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    KArgs& a = fresh_args();          // shadows the by-value copy inside this phase
    const auto& Gd = a.g[P.gi];
    if (P.publish) {
      // stream-K hand-off, producer side (cdna_hip_programming.md Guideline 16): plain stores, every wave drains its
      // stores, workgroup barrier, ONE agent-scope release, then the flag
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          a.sk_slab + (size_t)wl * (TM * TN * 16 * 256), 0, TM * TN * 16 * 256 * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            buf_store(sr, v, (unsigned)tid * 16u, (unsigned)((i * TN + j) * 4 + q) * 4096u);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.sk_flags + wl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (a.ksplit > 1) {
      // split-K: raw partial sums go to slab[group][split][m][col]; gg_reduce_kernel adds the splits in a
      // fixed order and applies bias / activation / epilogue, so the result does not depend on scheduling.
      float* slab = a.slab + Gd.slab_off + (size_t)P.split * a.M * Gd.Ncol;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0d + (wn * TN + j) * 32 + l31;
        if (col >= Gd.Ncol) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0d + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < a.M) slab[(size_t)m * Gd.Ncol + col] = acc[i][j][r];
          }
      }
    } else if constexpr (FUSE2) {
      // ---- ResidualBlock tail in the same launch: y = res + W2 . relu(W1 * x + b1) + b2 (reference common/elic.py:41-68).
      // Phase 1 above accumulated the 3x3 convolution TRANSPOSED (weights as the MFMA's A operand): lane l holds pixel
      // l % 32 of the wave's 32 rows, register r of tile j the channel 32 j + (r & 3) + 8 (r >> 2) + 4 (l / 32) -- which
      // is exactly how a lane feeds the A operand of v_mfma_f32_32x32x2_f32 (row = l % 32, k = l / 32), in exactly the k
      // order of the stand-alone 1x1 kernel's fragments (quad 2 g + h of a 16-deep stage).  So relu(acc + b1) goes into
      // the second contraction straight from the registers: no round trip through HBM or LDS, the same fma chains.
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 b1 = {0.f, 0.f, 0.f, 0.f};
          if (a.bias) b1 = *reinterpret_cast<const f32x4*>(a.bias + 32 * j + 8 * q + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[0][j][4 * q + e] = apply_act(acc[0][j][4 * q + e] + b1[e], a.act);
        }
      // One 32-wide column tile of the output at a time (16 accumulator registers next to the 48 that hold the A operand).
      // Its slice of W2 -- packed by the host in fragment order [Q][lane][4], 12 KB -- goes through one of two buffers in the
      // idle stage ring; the next slice and this tile's residual operand are loaded while the 48 MFMAs of the tile run, and
      // the tile leaves through a wave-private 4 KB staging slice behind the two buffers.
      float* stage = ring + 2 * 3072 + wave * 1024;
      const f32x4* w2src = reinterpret_cast<const f32x4*>(a.w2f);
      f32x4 wreg[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) wreg[u] = w2src[te + 256 * u];
      const int c4 = (lane & 7) << 2;                // epilogue: 8 lanes x 16 B per row, 8 rows per pass, 4 passes
      const int rsub = lane >> 3;
      const int mrow = m0d + wave * 32 + rsub;       // + 8 p: this lane's output row in pass p
      __syncthreads();                               // the K loop's last fragment reads are done: the ring is free
#pragma unroll 1
      for (int ct = 0; ct < (SNTC_DBG(a, 128) ? 0 : 6); ++ct) {   // output channels 32 ct ... 32 ct + 31
        float* w2s = ring + (ct & 1) * 3072;
#pragma unroll
        for (int u = 0; u < 3; ++u) reinterpret_cast<f32x4*>(w2s)[te + 256 * u] = wreg[u];
        __syncthreads();                             // (the other buffer's readers are a tile behind this barrier)
        if (ct < 5) {
#pragma unroll
          for (int u = 0; u < 3; ++u) wreg[u] = w2src[(ct + 1) * 768 + te + 256 * u];
        }
        const int ch = 32 * ct + c4;
        f32x4 rv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          rv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (a.epi != SNTC_EPI_STORE && mrow + 8 * p < a.M)
            rv[p] = *reinterpret_cast<const f32x4*>(a.res + (size_t)(mrow + 8 * p) * a.Cout2 + ch);
        }
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(w2s + ((4 * j + q) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[0][j][4 * q + e], bq[e], acc2, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc2[r];
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias2) bv = *reinterpret_cast<const f32x4*>(a.bias2 + ch);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int m = mrow + 8 * p;
          if (m >= a.M) continue;
          const size_t idx = (size_t)m * a.Cout2 + ch;         // forward convolution: the output pixel index is the row index
          f32x4 v = *reinterpret_cast<const f32x4*>(stage + (rsub + 8 * p) * 32 + c4) + bv;
          if (a.epi != SNTC_EPI_STORE) v = apply_epilogue(v, a.epi, rv[p], a.aux, idx);
          *reinterpret_cast<f32x4*>(a.y + idx) = v;
        }
      }
    } else if ((a.Cout & 3) == 0) {
      // Wide path (Cout % 4 == 0): each wave transposes its accumulators through a private LDS slice (the stage
      // ring is idle after the last barrier) so that every lane owns 4 consecutive channels of one pixel: bias /
      // residual / gate operands are read and the output is written with 16-B accesses.
      float* stage = ring + (DMA ? RING * SLOT - 4 * EPW : 0) + wave * EPW;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j0 = 0; j0 < TN; j0 += 2) {
        const int ct = (TN - j0) >= 2 ? 2 : 1;          // tiles in this chunk
        const int wfl = ct * 32;                         // chunk width in floats
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (jj < ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              stage[((r & 3) + 8 * (r >> 2) + 4 * h) * wfl + jj * 32 + l31] = acc[i][j0 + jj][r];
          }
        }
        const int lanes_per_row = wfl >> 2;              // 16 or 8
        const int rows_per_pass = 64 / lanes_per_row;    // 4 or 8
        const int c4 = (lane % lanes_per_row) << 2;
        const int rsub = lane / lanes_per_row;
        const int col = n0d + (wn * TN + j0) * 32 + c4;
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
        for (int rp = 0; rp < 32; rp += rows_per_pass) {
          const int rloc = rp + rsub;
          const int4 ri = rinfo[(wm * TM + i) * 32 + rloc];
          const int oy = ri.y * a.sO + oyo;
          const int ox = ri.z * a.sO + oxo;
          if (!col_ok || !ri.w || (unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
          const size_t idx = (((size_t)ri.x * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
          f32x4 v = *reinterpret_cast<const f32x4*>(stage + rloc * wfl + c4) + bv;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
          if (a.epi != SNTC_EPI_STORE)
            v = apply_epilogue(v, a.epi, SNTC_DBG(a, 32) ? v : *reinterpret_cast<const f32x4*>(a.res + idx), a.aux, idx);
          if (!SNTC_DBG(a, 16) || v[0] == 12345.678f) *reinterpret_cast<f32x4*>(a.y + idx) = v;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0d + (wn * TN + j) * 32 + l31;
        if (col >= Gd.Ncol) continue;
        const unsigned ce = Gd.cols[col];
        const int ch = ce & 0xffff;
        const int oyo = (int)((ce >> 24) & 0xff) - 128;
        const int oxo = (int)((ce >> 16) & 0xff) - 128;
        const float bv = a.bias ? a.bias[ch] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int4 ri = rinfo[row];
          if (!ri.w) continue;
          const int oy = ri.y * a.sO + oyo;
          const int ox = ri.z * a.sO + oxo;
          if ((unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
          const size_t idx = (((size_t)ri.x * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
          a.y[idx] = apply_epilogue1(apply_act(acc[i][j][r] + bv, a.act), a.epi, a.res, a.aux, idx);
        }
      }
    }
    if (!more) break;
    __syncthreads();          // epilogue staging reads done before the next piece's stages land in the ring
    P = Q;
    rb = qb;
  }
}

// ---- the instantiations the library ships, one translation unit per mode (a new mode gets a file of its own: the fp32
// instances that carry the decode are then not even recompiled, and tools/kernel_resources.py pins their register allocation).
//   X(TM, TN, WM, WN): tile variants 1..7 (128 x 32 v), 8 (64 x 64), 9 (128 x 128), 10 (256 x 128)
#define SNTC_GG_SHAPES(X) X(1, 1, 4, 1) X(1, 2, 4, 1) X(1, 3, 4, 1) X(1, 4, 4, 1) X(1, 5, 4, 1) X(1, 6, 4, 1) X(1, 7, 4, 1) X(1, 1, 2, 2) X(2, 2, 2, 2) X(2, 4, 4, 1)
#define SNTC_GG_DMA_SHAPES(X) X(1, 1, 4, 1) X(1, 2, 4, 1) X(1, 3, 4, 1) X(1, 4, 4, 1) X(1, 5, 4, 1) X(1, 1, 2, 2) X(2, 2, 2, 2)
constexpr int kDeepRing = 8;   // ring slots of the deep direct-to-LDS instance (64 x 64)

}  // namespace sntc

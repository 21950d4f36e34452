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
//    once stage j+1 has landed (counted vmcnt) -- one barrier per 16-deep stage;
//  * HALO instances (stride-1 sampling on a macro grid equal to the input grid: the 3x3 / 1 layer, the phase groups of the
//    stride-2 transposed layers): the T taps of one 16-channel slab read the SAME activation rows shifted by whole pixels, so
//    the workgroup stages ONE patch of BM + (th - 1) W + tw - 1 consecutive pixels per slab (two or three patch buffers) and
//    only the weights per stage -- 142 KB instead of 324 KB per nine stages of the 480 -> 640 layer; a fragment row of tap
//    (ty, tx) is patch row r + sy W + sx, zeroed in registers where the tap leaves the image (the loop then reaches the LDS
//    read ceiling of the microbench: 280 against 232 TFLOP/s-equivalent at 256 x 128).
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
template <int WM, int WN, int TM, int TN, bool HALO, bool DBUF>
__global__ void __launch_bounds__(WM* WN * 64, 2) bf3_kernel(const GGArgs a) {
  constexpr int NT = WM * WN * 64;
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int P_R = kBf3PatchRounds;                    // HALO: a patch is at most P_R chunk rounds (426 rows at 512 threads)
  constexpr int PAREA = HALO ? 2 * P_R * NT * 16 : 0;     // HALO: the patch buffers (two of <= P_R rounds or three of <= 3), 80 KB
  constexpr int SLOT = (HALO ? BN : BM + BN) * 96;        // bytes per ring slot (HALO: weights only)
  constexpr int NS = (HALO && DBUF) ? kBf3DeepRing : 3;   // ring slots
  constexpr int A_CH = BM * 6 / NT;                       // 16-B chunks of A per thread and stage
  constexpr int B_CH = (BN * 6 + NT - 1) / NT;            // of B (the last round may cover only the first waves)
  static_assert(BM * 6 % NT == 0, "A chunks must divide evenly over the threads");
  constexpr int EPW = 32 * 32;                            // floats of epilogue staging per wave (one 32 x 32 accumulator tile)
  static_assert(HALO || SLOT >= NW * EPW * 4, "epilogue staging must fit in ring slot 2 (slots 0 and 1 take the next piece's first stages)");
  static_assert(!HALO || PAREA - P_R * NT * 16 >= NW * EPW * 4, "epilogue staging must fit behind the next piece's first patch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem + PAREA;
  int4* rinfo_all = reinterpret_cast<int4*>(smem + PAREA + NS * SLOT);     // [2][BM] (n, qy, qx, valid) of the current / next tile
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
    // Unit order (a.order == 1, the default): row strip major, then group, then column tile -- every worker's share mixes the
    // groups (their tiles differ in length: a group-major order hands some workers only short, epilogue-heavy tiles).
    // a.order == 0: COLUMN tile outermost, row strips inside (u = ntm * unit0(g) + nt * ntm * steps + mt * steps + k): a tile's
    // weight rows stay in one XCD's L2 while the strips stream past; measured equal on single-group layers and 20 % slower on
    // the four-group synthesis, kept for the A/B.
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
  constexpr int A_N = HALO ? 1 : A_CH;
  int a_row[A_N];
  unsigned a_part[A_N];               // byte offset of the lane's chunk inside the 96-B block of a (pixel, slab)
  int a_iy0[A_N], a_ix0[A_N];
  unsigned a_img[A_N], a_off[A_N];
  unsigned b_off[B_CH];
  if (!HALO) {
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int q = i * NT + tid, row = q / 6, part = q - row * 6;
      a_row[i] = row;
      a_part[i] = (unsigned)((part >> 1) * 32 + (((part & 1) ^ ((row >> 3) & 1)) << 4));
    }
  }
  int ld_stage = 0, ld_t = 0, ld_cc = 0, ld_ty = 0, ld_tx = 0, g_T = 1, g_tw = 1;
  int n0 = 0;
  const int my_b = [&]() {                       // B instructions this wave issues per stage
    int nb = 0;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) nb += (i * NT + wave * 64 < BN * 6) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(nb);
  }();
  // HALO: the patch of one channel slab = pixels [tile's first pixel + dmin, ... + BM + span) of the flattened [N H W] input
  constexpr int P_N = HALO ? P_R : 1;
  unsigned p_off[P_N];                 // per chunk round: byte offset of the lane's chunk at slab 0, or out of range
  int h_rounds = 0, h_pbytes = 0, h_nbufs = 2, h_th = 1;
  int h_shift0 = 0, h_coljump = 1, h_rowjump = 1, h_zero96 = 0;   // tap walk in patch rows; byte offset of the patch's zero row
  int rd_t = 0, rd_tx = 0, rd_shift = 0, rd_buf = 0;              // read pointer: tap, its column, its row shift, its slab's buffer
  int r_row[TM];                       // the lane's fragment rows inside the tile
  unsigned r_bits[TM];                 // bit t: tap t of the current group stays inside the image for that row
#pragma unroll
  for (int i = 0; i < TM; ++i) r_row[i] = (wm * TM + i) * 32 + l31;
  int issued = 0;                      // DMA instructions this wave has issued for the current piece (the marks below count in it)

  auto set_tap = [&](int ty, int tx) {
    if (HALO) return;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
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
    if (HALO) {
      g_T = __builtin_amdgcn_readfirstlane(g_T);
      g_tw = __builtin_amdgcn_readfirstlane(g_tw);
      h_th = __builtin_amdgcn_readfirstlane(g_T / g_tw);
      // Rows are macro pixels (n, qy, qx) of the [Qh, Qw] grid, flattened; with sA = 1 tap (ty, tx) of row m reads the input
      // pixel at macro position m + (q0y + offy + ty tstep) Qw + q0x + offx + tx tstep -- when that position lies inside the
      // image ([H, W] <= [Qh, Qw]; the grids are equal except for the transposed layers whose phases do not share one origin).
      const int W = a.W, H = a.H, ts = a.tstep, Qw = a.Qw;
      const int prows = BM + (h_th - 1) * Qw + g_tw - 1;
      // a row whose tap leaves the image reads the zero row behind the patch; a 1-tap group keeps its patch at 3 rounds (three
      // buffers: its patch changes every stage) and masks the data instead (-1), or not at all if it is a plain 1x1 (-2)
      const int zrow = g_T > 1 ? 1 : 0;
      h_rounds = __builtin_amdgcn_readfirstlane(((prows + zrow) * 6 + NT - 1) / NT);      // <= P_R (host check)
      h_pbytes = h_rounds * NT * 16;
      h_nbufs = h_rounds * 3 <= 2 * P_R ? 3 : 2;
      const bool plain = G.q0y == 0 && G.q0x == 0 && a.offy == 0 && a.offx == 0 && a.Qh == a.H && Qw == a.W;
      h_zero96 = g_T > 1 ? prows * 96 : plain ? -2 : -1;
      h_coljump = ts;
      h_rowjump = ts * (Qw - g_tw + 1);
      h_shift0 = ts < 0 ? (h_th - 1) * Qw + g_tw - 1 : 0;
      const int dmin = (G.q0y + a.offy - (ts < 0 ? h_th - 1 : 0)) * Qw + G.q0x + a.offx - (ts < 0 ? g_tw - 1 : 0);
      const int gbase = p.mt * BM + dmin;
      const bool same_grid = a.Qh == H && Qw == W;
      const unsigned total = (unsigned)a.M;
#pragma unroll
      for (int i = 0; i < P_N; ++i) {
        const int q = i * NT + tid, row = q / 6, part = q - row * 6;
        const int gp = gbase + row;
        bool ok = row < prows && (unsigned)gp < total;
        unsigned pix = (unsigned)gp;
        if (!same_grid && ok) {                                  // macro position -> input pixel, if it has one
          const int per = a.Qh * Qw;
          const int n = gp / per, rem = gp - n * per;
          const int y = rem / Qw, x = rem - y * Qw;
          ok = y < H && x < W;
          pix = (unsigned)((n * H + y) * W + x);
        }
        p_off[i] = ok ? pix * (unsigned)a.Cin * 6u + (unsigned)((part >> 1) * 32 + (((part & 1) ^ ((row >> 3) & 1)) << 4)) : kOOR;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int4 ri = rinfo[r_row[i]];
        unsigned bits = 0;
        if (h_zero96 == -2) {
          bits = 1u;                        // rows past M read rows past the input: zero-filled by the DMA
        } else if (ri.w) {
          int t = 0;
          for (int ty = 0; ty < h_th; ++ty)
            for (int tx = 0; tx < g_tw; ++tx, ++t) {
              const bool ok = (unsigned)(ri.y + a.offy + ty * ts) < (unsigned)H && (unsigned)(ri.z + a.offx + tx * ts) < (unsigned)W;
              bits |= (ok ? 1u : 0u) << t;
            }
        }
        r_bits[i] = bits;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_N; ++i) {
        const int4 ri = rinfo[a_row[i]];
        a_iy0[i] = ri.y * a.sA + a.offy;
        a_ix0[i] = ri.z * a.sA + a.offx;
        a_img[i] = ri.w ? (unsigned)ri.x * (unsigned)(a.H * a.W) * (unsigned)a.Cin * 6u + a_part[i] : kOOR;
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {   // rows past Ncol re-read the last column (finite, discarded)
      const int q = i * NT + tid, row = q / 6, part = q - row * 6;
      const int brow = min(n0 + row, G.Ncol - 1);
      b_off[i] = (unsigned)brow * (unsigned)(G.K / kStage) * 96u + (unsigned)((part >> 1) * 32 + (((part & 1) ^ ((row >> 3) & 1)) << 4));
    }
    ld_stage = __builtin_amdgcn_readfirstlane(p.k0);
    ld_cc = __builtin_amdgcn_readfirstlane(p.k0 / g_T);
    ld_t = ld_stage - ld_cc * g_T;
    ld_ty = __builtin_amdgcn_readfirstlane(ld_t / g_tw);
    ld_tx = ld_t - ld_ty * g_tw;
    set_tap(ld_ty, ld_tx);
  };
  auto issue_b = [&](int slot) {                // this wave's share of one stage of weights
    const unsigned wsoff = (unsigned)__builtin_amdgcn_readfirstlane(ld_stage) * 96u;
    char* bb = ring + slot * SLOT + wave * 1024 + (HALO ? 0 : BM * 96);
    if (!SNTC_DBG(a, 1) || issued < 40) {      // (diagnostic: after the first stages the LDS keeps stale but real data)
#pragma unroll
      for (int i = 0; i < B_CH; ++i)
        if ((BN * 6) % NT == 0 || i * NT + wave * 64 < BN * 6)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)(bb + i * NT * 16), 16, (int)b_off[i], (int)wsoff, 0, 0);
    }
    issued += my_b;
  };
  auto issue = [&](int slot) {                  // this wave's share of one stage, L2 / HBM -> LDS
    char* base = ring + slot * SLOT + wave * 1024;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(ld_cc) * 96u;
    if (!HALO) {
#pragma unroll
      for (int i = 0; i < A_N; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)(base + i * NT * 16), 16, (int)a_off[i], (int)soff, 0, 0);
    }
    issue_b(slot);
    // next stage: channel slab outermost, taps inside (k = cc * T * 16 + t * 16 + c), branch-free
    ++ld_stage;
    if (!HALO) {
      const int row_end = (ld_tx + 1 == g_tw) ? 1 : 0;
      const int tap_end = (ld_t + 1 == g_T) ? 1 : 0;
      ld_tx = row_end ? 0 : ld_tx + 1;
      ld_ty = tap_end ? 0 : ld_ty + row_end;
      ld_t = tap_end ? 0 : ld_t + 1;
      ld_cc += tap_end;
      set_tap(ld_ty, ld_tx);
    }
  };
  auto issue_patch = [&](int cc, int buf) {     // HALO: this wave's share of the patch of channel slab cc
    char* base = smem + buf * h_pbytes + wave * 1024;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(cc) * 96u;
#pragma unroll
    for (int i = 0; i < P_N; ++i)
      if (i < h_rounds && (!SNTC_DBG(a, 2) || issued < 40)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)(base + i * NT * 16), 16, (int)p_off[i], (int)soff, 0, 0);
    issued += h_rounds;
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
  // HALO: leave at most n of this wave's DMA instructions in flight (n is wave-uniform)
  // (a jump into a table of s_waitcnt: the compiler turns a switch / an if-tree over thirteen asm statements into ~50 scalar
  // instructions of flag juggling per call, and this runs once per 16-deep stage)
  auto wait_vm = [&](int n) {
    const int off = min(n, 12) * 8 + 12;       // table entry = s_waitcnt + s_branch; 12 bytes from the s_getpc result to the table
    asm volatile(
        "s_getpc_b64 s[94:95]\n"
        "s_add_u32 s94, s94, %0\n"
        "s_addc_u32 s95, s95, 0\n"
        "s_setpc_b64 s[94:95]\n"
        "s_waitcnt vmcnt(0)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(1)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(2)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(3)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(4)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(5)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(6)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(7)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(8)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(9)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(10)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(11)\n s_branch .Lvm_done%=\n"
        "s_waitcnt vmcnt(12)\n"
        ".Lvm_done%=:\n"
        :
        : "s"(off)
        : "memory", "s94", "s95", "scc");
  };

  // ---------------------------------------------------------------- fragments + MFMA
  const int hoff = (h ^ ((l31 >> 3) & 1)) << 4;
  const int fa = (wm * TM * 32 + l31) * 96 + hoff;
  const int fb = ((HALO ? 0 : BM) + wn * TN * 32 + l31) * 96 + hoff;
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
  // HALO: the fragments of the stage the read pointer (rd_buf, rd_t, rd_shift) stands on.  A row of tap t is patch row
  // r + shift(t); where the tap leaves the image for that row (bit t of r_bits clear) the lane reads the patch's zero row instead
  auto read_frag_halo = [&](Frag& F, int slot) {
    const int pbo = rd_buf * h_pbytes;
    const char* base = ring + slot * SLOT;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int prow = r_row[i] + rd_shift;
      int off = prow * 96 + ((h ^ ((prow >> 3) & 1)) << 4);
      const bool in = (r_bits[i] >> rd_t) & 1u;
      off = (in || h_zero96 < 0) ? off : h_zero96;
      const char* src = smem + pbo + off;
#pragma unroll
      for (int p = 0; p < 3; ++p) F.a[p][i] = *reinterpret_cast<const bf16x8*>(src + p * 32);
      if (h_zero96 == -1) {                 // 1-tap group with an offset origin: no zero row, mask the data
        const unsigned keep = in ? 0xffffffffu : 0u;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x4 v = __builtin_bit_cast(u32x4, F.a[p][i]);
          v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
          F.a[p][i] = __builtin_bit_cast(bf16x8, v);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < TN; ++j) F.b[p][j] = *reinterpret_cast<const bf16x8*>(base + fb + j * 32 * 96 + p * 32);
  };
  auto advance_read = [&](bool slab_end) {
    if (slab_end) {
      rd_t = 0; rd_tx = 0; rd_shift = h_shift0;
      rd_buf = rd_buf + 1 == h_nbufs ? 0 : rd_buf + 1;
    } else {
      ++rd_t;
      if (++rd_tx == g_tw) { rd_tx = 0; rd_shift += h_rowjump; }
      else rd_shift += h_coljump;
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

  // HALO pipeline state.  Marks are values of `issued` right after a group of loads went out; a load has landed once at most
  // issued - mark younger instructions are in flight (vector memory returns in order).
  int mBc = 0, mBn = 0, mBn2 = 0;      // weights of the current / next / next-but-one stage
  int mPc = 0, mP1 = 0, mP2 = 0;       // patch of the current / next / next-but-one slab
  int nP = 0, nP_buf = 0, c_last = 0;  // next slab to stage, its buffer, the piece's last slab
  int pm[NS - 1];                      // marks of the weights the prologue issued (stages 0 .. NS - 2)
  // first loads of a piece: (HALO) the patches of its first slabs, then the first NS - 1 stages' weights (ring slots 0 ...)
  auto start_piece = [&](const Piece& p, int rb) {
    init_loader(p, rb);
    const int n = __builtin_amdgcn_readfirstlane(p.k1 - p.k0);
    if (HALO) {
      issued = 0;
      const int c0 = ld_cc;
      c_last = __builtin_amdgcn_readfirstlane((p.k1 - 1) / g_T);
      rd_t = ld_t; rd_tx = ld_tx; rd_buf = 0;
      rd_shift = h_shift0 + ld_ty * (h_rowjump + (g_tw - 1) * h_coljump) + ld_tx * h_coljump;
      issue_patch(c0, 0);
      mPc = issued;
      if (h_nbufs == 3 && c0 + 1 <= c_last) issue_patch(c0 + 1, 1);
      mP1 = issued;
      mP2 = issued;
      nP = c0 + h_nbufs - 1;
      nP_buf = h_nbufs - 1;
#pragma unroll
      for (int q = 0; q < NS - 1; ++q) {                   // weights of the first NS - 1 stages
        if (q < n) issue(q);
        pm[q] = issued;
      }
      mBc = pm[0];
      mBn = pm[1];
    } else {
      if (n > 0) issue(0);
      if (n > 1) issue(1);
    }
  };

  // ---------------------------------------------------------------- the piece loop
  int rb = 0;
  write_rinfo(P, rb);
  __syncthreads();
  start_piece(P, rb);
  while (true) {
    const int n = __builtin_amdgcn_readfirstlane(P.k1 - P.k0);
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

    Frag F;
    int s_cur = 0, s_n1 = 1, s_n2 = 2;
    if constexpr (HALO && DBUF) {
      // Fragments double-buffered: step j multiplies stage j from registers while it READS stage j+1 and the DMA fills the
      // slot stage j left with stage j+3 (step -1 only reads).  A wave issues in order and its 24 MFMAs of a stage keep the
      // matrix pipe busy for 768 cycles: everything else a step does (DMA issue, fragment addresses and reads, tap masks, the
      // pipeline bookkeeping) is spread BETWEEN the six MFMA groups so that it issues in their shadow instead of after them
      // (with the bookkeeping behind the MFMAs the loop ran at 0.70 of the microbench's rate).
      // Stage 0's weights and its slab's patch must have landed first.  (Everything younger that is not counted in `issued`
      // -- the previous piece's epilogue stores, the hand-off loads above -- only makes a wait longer, never shorter.)
      Frag G;
      wait_vm(issued - max(mBc, mPc));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      int mq[NS - 2];                                      // marks of the weights of stages j+2 ... j+NS-1
#pragma unroll
      for (int q = 0; q < NS - 2; ++q) mq[q] = pm[q + 1];
      int s_rd = 0, s_is = NS - 1;                         // ring slot of the stage being read (j+1) / of stage j: free, takes stage j+NS
      // MFMA number q of a stage (q = term * TM * TN + i * TN + j; smallest terms first) and a scheduling fence behind it
      auto mf = [&](const Frag& Fc, bool on, int q) {      // q is a literal / an unrolled loop index: everything below folds
        const int t = q / (TM * TN), i = (q / TN) % TM, j2 = q % TN;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
        if (on && !SNTC_DBG(a, 64))
          acc[i][j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fc.a[PA[t]][i], Fc.b[PB[t]][j2], acc[i][j2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto step = [&](auto EDGE_, int j, Frag& Fc, Frag& Fn) {
        constexpr bool EDGE = decltype(EDGE_)::value;      // first / last steps of a piece: some of the parts below are absent
        const bool mm = !EDGE || j >= 0, rd = (!EDGE || j + 1 < n) && (!SNTC_DBG(a, 8) || j < 1);
        // An MFMA keeps the matrix pipe busy for 32 cycles and the SIMD's vector issue for 8 of them; what fits beside it is a
        // handful of VALU / scalar instructions, or ONE ds_read_b128 (two waves per SIMD: the LDS array takes two per gap and
        // SIMD before it, not the MFMA, sets the pace), or one DMA piece.  So the step's other work goes out in slices of that
        // size, one behind each MFMA: fragment addresses and the twelve fragment reads of stage j+1 first, then the DMA
        // pieces of stage j+NS (and of the next patch), then the pipeline bookkeeping.
        static_assert(TM == 2 && TN == 2, "the slices below are laid out for a 64 x 64 wave tile (24 MFMAs per stage)");
        const char* src[TM];
        bool in[TM];
        const char* bsrc = ring + s_rd * SLOT + fb;
        int q = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int prow = r_row[i] + rd_shift;
          int off = prow * 96 + ((h ^ ((prow >> 3) & 1)) << 4);
          in[i] = (r_bits[i] >> rd_t) & 1u;
          off = (in[i] || h_zero96 < 0) ? off : h_zero96;
          src[i] = smem + rd_buf * h_pbytes + off;
          __builtin_amdgcn_sched_barrier(0);
          mf(Fc, mm, q++);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            if (rd) Fn.a[p][i] = *reinterpret_cast<const bf16x8*>(src[i] + p * 32);
            __builtin_amdgcn_sched_barrier(0);
            mf(Fc, mm, q++);
          }
        }
#pragma unroll
        for (int j2 = 0; j2 < TN; ++j2)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            if (rd) Fn.b[p][j2] = *reinterpret_cast<const bf16x8*>(bsrc + j2 * 32 * 96 + p * 32);
            __builtin_amdgcn_sched_barrier(0);
            mf(Fc, mm, q++);
          }
        // q == 14: weights of stage j+NS into the slot stage j left
        if (!EDGE || j + NS < n) issue(s_is);
        const int mnew = issued;
        __builtin_amdgcn_sched_barrier(0);
        mf(Fc, mm, q++);
        if (rd && ((EDGE && j < 0) || rd_t == 0) && nP <= c_last) {   // reading a slab's first stage: its predecessor's buffer is free
          issue_patch(nP, nP_buf);
          if (h_nbufs == 3) mP2 = issued;
          else mP1 = issued;
          ++nP;
          nP_buf = nP_buf + 1 == h_nbufs ? 0 : nP_buf + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        mf(Fc, mm, q++);
        const bool slab_end = rd && rd_t + 1 == g_T;
        const int need = slab_end ? max(mq[0], mP1) : mq[0];
        const int allowed = issued - need;
        __builtin_amdgcn_sched_barrier(0);
        mf(Fc, mm, q++);
        s_rd = s_rd + 1 == NS ? 0 : s_rd + 1;
        s_is = s_is + 1 == NS ? 0 : s_is + 1;
#pragma unroll
        for (int k = 0; k + 1 < NS - 2; ++k) mq[k] = mq[k + 1];
        mq[NS - 3] = mnew;
        __builtin_amdgcn_sched_barrier(0);
        mf(Fc, mm, q++);
        if (rd) {
          advance_read(slab_end);
          if (slab_end) mP1 = mP2;
        }
        __builtin_amdgcn_sched_barrier(0);
        mf(Fc, mm, q++);
        if (rd && h_zero96 == -1) {         // 1-tap group with an offset origin: no zero row, mask the data
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const unsigned keep = in[i] ? 0xffffffffu : 0u;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              u32x4 v = __builtin_bit_cast(u32x4, Fn.a[p][i]);
              v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
              Fn.a[p][i] = __builtin_bit_cast(bf16x8, v);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (; q < TM * TN * 6; ++q) mf(Fc, mm, q);
        if (!EDGE || j + 2 < n) {
          // stage j+2's weights -- and its patch, if it opens the next slab -- have landed; younger loads may still fly
          wait_vm(allowed);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!SNTC_DBG(a, 4)) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      using Edge = std::integral_constant<bool, true>;
      using Fast = std::integral_constant<bool, false>;
      int j = -1;
      step(Edge{}, j, G, F);                               // reads stage 0 into F
      for (j = 0; j + NS + 1 < n; j += 2) {
        step(Fast{}, j, F, G);
        step(Fast{}, j + 1, G, F);
      }
      for (; j < n; j += 2) {
        step(Edge{}, j, F, G);
        if (j + 1 < n) step(Edge{}, j + 1, G, F);
      }
    } else if constexpr (HALO) {
      // stage 0's weights and its slab's patch must have landed.  (Everything younger that is not counted in `issued` -- the
      // previous piece's epilogue stores, the hand-off loads above -- only makes a wait longer, never shorter.)
      wait_vm(issued - max(mBc, mPc));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      for (int j = 0; j < n; ++j) {
        if (j + 2 < n) {                                   // weights of stage j+2 into the slot stage j-1 left at the last barrier
          issue(s_n2);
          mBn2 = issued;
        }
        if ((j == 0 || rd_t == 0) && nP <= c_last) {       // first stage of a slab: the patch of slab + (buffers - 1) into the
          issue_patch(nP, nP_buf);                         // buffer the previous slab left at the last barrier
          if (h_nbufs == 3) mP2 = issued;
          else mP1 = issued;
          ++nP;
          nP_buf = nP_buf + 1 == h_nbufs ? 0 : nP_buf + 1;
        }
        read_frag_halo(F, s_cur);
        mfma6(F);
        __builtin_amdgcn_sched_barrier(0);
        const bool slab_end = rd_t + 1 == g_T;
        if (j + 1 < n) {
          // stage j+1's weights -- and its patch, if it opens the next slab -- have landed; younger loads may still fly
          const int need = slab_end ? max(mBn, mP1) : mBn;
          wait_vm(issued - need);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        const int t = s_cur; s_cur = s_n1; s_n1 = s_n2; s_n2 = t;
        mBn = mBn2;
        advance_read(slab_end);
        if (slab_end) { mPc = mP1; mP1 = mP2; }
      }
    } else {
      // stage 0 must have landed before the first step; stage 1 may still fly.  (Everything older than the two stage issues --
      // the previous piece's epilogue stores were issued AFTER them -- only makes this wait longer, never shorter.)
      if (n > 1) wait_stages(I1{});
      else wait_stages(I0{});
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);

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
    }

    // ---- the next piece's first two stages go in flight (ring slots 0 and 1) before this piece's results are stored
    Piece Q;
    const bool more = next_piece(&Q);
    const int n0d = n0;
    const int4* rinfo = rinfo_all + rb * BM;
    __syncthreads();                  // every wave has read its last fragments: the whole ring is free
    if (more) {
      write_rinfo(Q, rb ^ 1);
      __syncthreads();
      start_piece(Q, rb ^ 1);
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
      // (HALO: of the tail of the patch area, behind the next piece's first patches)
      float* stage = reinterpret_cast<float*>(HALO ? smem + PAREA - NW * EPW * 4 : ring + 2 * SLOT) + wave * EPW;
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
// variants + launch (ids continue csrc/gather_gemm.hip's: 11 = 256 x 256, 12 = 256 x 128, 13 = 256 x 192 -- the exact fit of the
// N = 192 layers, which lose a quarter of either of the others to padding; all 512 threads, one workgroup per CU)
// ---------------------------------------------------------------------------------------------
static const void* bf3p_kernel(int v, bool halo) {
  switch (v) {
    case 11: return halo ? reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 4, true, false>) : reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 4, false, false>);
    case 12: return halo ? reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 2, true, true>) : reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 2, false, false>);
    case 13: return halo ? reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 3, true, false>) : reinterpret_cast<const void*>(&bf3_kernel<4, 2, 2, 3, false, false>);
    default: return nullptr;
  }
}

int bf3p_variant_bm(int v) { return 256; }
int bf3p_variant_bn(int v) { return v == 11 ? 256 : v == 13 ? 192 : 128; }
size_t bf3p_sk_slab_floats(int v) { return (size_t)(v == 11 ? 8 : v == 13 ? 6 : 4) * 16 * 512; }

static size_t bf3p_lds_bytes(int v, bool halo) {
  const size_t rinfo = 2 * bf3p_variant_bm(v) * sizeof(int4);
  if (halo) return (size_t)2 * kBf3PatchRounds * 512 * 16 + (size_t)(v == 12 ? kBf3DeepRing : 3) * bf3p_variant_bn(v) * 96 + rinfo;
  return (size_t)3 * (bf3p_variant_bm(v) + bf3p_variant_bn(v)) * 96 + rinfo;
}
int bf3p_patch_rows_max() { return kBf3PatchRounds * 512 / 6; }

static std::once_flag g_bf3p_once[16];
static int g_bf3p_rc[16];

int bf3p_init() {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the residency tables");
  std::call_once(g_bf3p_once[dev], [&] {
    g_bf3p_rc[dev] = SNTC_OK;
    for (int v : {11, 12, 13})
      for (bool halo : {false, true}) {
        hipError_t e = hipFuncSetAttribute(bf3p_kernel(v, halo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf3p_lds_bytes(v, halo));
        if (e != hipSuccess) g_bf3p_rc[dev] = hip_fail(e, "bf3p_init");
      }
  });
  return g_bf3p_rc[dev];
}

int bf3p_launch(int variant, const GGArgs& args, int nblocks, hipStream_t stream) {
  const bool halo = args.halo != 0;
  const void* fn = bf3p_kernel(variant, halo);
  if (!fn) return fail(SNTC_ERR_UNSUPPORTED, "unknown pre-split bf16 x 3 tile variant");
  int rc = bf3p_init();
  if (rc) return rc;
  GGArgs a = args;
#ifdef SNTC_DIAG
  if (const char* e = getenv("SNTC_GG_DBG")) a.dbg = atoi(e);   // diagnostic builds only (make DIAG=1): results are WRONG with it
#endif
  void* params[] = {&a};
  hipError_t e = hipLaunchKernel(fn, dim3(nblocks), dim3(512), params, bf3p_lds_bytes(variant, halo), stream);
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

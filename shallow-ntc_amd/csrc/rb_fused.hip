// rb_fused.hip -- the ELIC ResidualBlock (reference common/elic.py:41-68) as ONE launch on a 2-D pixel tile:
//
//     y = x + conv1x1_{c/2 -> c}( relu(conv3x3_{c/2 -> c/2}( relu(conv1x1_{c -> c/2}(x)) )) )
//
// A workgroup (512 threads = 8 waves, one per CU: the tile's patch fills the LDS) owns a tile of 8 rows x 32 pixels:
//   head   the 1x1 head on the (8+2) x (32+2) HALO patch, accumulated TRANSPOSED (weights = MFMA A operand, pixels = B
//          operand read straight from global memory, 16 B per lane): lane l then holds pixel l % 32 and four consecutive
//          channels per register quad -- one 16-B chunk of the patch's LDS image [16-channel slab][patch pixel][16], so
//          relu(acc + b0) goes to LDS with ds_write_b128 and out-of-image pixels become the 3x3's zero padding;
//   3x3    nine taps = nine SHIFTED fragment reads of that one patch (32 consecutive patch pixels per fragment: the
//          XOR swizzle stays conflict-free under any shift) -- no per-tap global gathers, no staging writes; wave w owns
//          tile row w and all c/2 output channels (lane = pixel, registers = channels);
//   tail   relu(acc + b1) is, as it stands in the registers, the B operand of the 1x1 tail (k = lane half, channel quads
//          in the fragment order of a 16-deep stage); the result leaves with bias, the skip (+ x) and 16-B stores.
// The three weight sets travel as ONE stream of 6-KB units (a unit = the LDS image of one 16-deep K stage of all c/2
// output rows) through a three-slot ring filled by LDS-DMA (buffer_load_dwordx4 ... lds, 1-KB pieces); the
// workgroup is persistent and the stream wraps from one tile's tail into the next tile's head.  Tiles are walked down the
// image and the patch is a ring of rows: a tile below the previous one re-uses its two top halo rows and computes 8 new ones.
//
// Every output element is the SAME k-ordered fp32 fma chain as in the three stand-alone gather-GEMM launches
// (gather_gemm.hip: stage = 16 channels, MFMA e of k-group g sums k in {8g+e, 8g+4+e}; 3x3: channel slab outermost,
// taps inside; products commute, zero padding contributes fma(0, w, acc)), so results are bit-identical to them.
#include <algorithm>
#include <mutex>
#include <type_traits>
#include "sntc_internal.h"

#include "rb_common.h"

namespace sntc {
using namespace rb;

namespace {

#define RB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int C>
__global__ void __launch_bounds__(512, 2) rb_kernel(const RBArgs a) {
  using K = RBCfg<C>;
  constexpr int CH = K::CH, NT = K::NT, SL = K::SL, PW = K::PW, PP = K::PP, UNIT = K::UNIT;
  constexpr int U0 = K::U0, U1 = K::U1, UT = K::UT, RING = K::RING;
  typedef __attribute__((address_space(3))) void lds_void;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);        // [SL][PP][16], 16-B chunks XOR-swizzled by (pixel >> 2) & 3
  float* ring = patch + K::PATCH;                       // [RING][CH][16], the packed units as they are
  float* lbias = ring + RING * UNIT;                    // b0 | b1 | b2

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // this workgroup's contiguous range of tiles; workgroups b, b + 8, ... (one XCD under round-robin placement) own one
  // contiguous eighth of the launch, so the halo rows neighbouring tiles share are L2 hits (speed only)
  const int G = gridDim.x, b = blockIdx.x;
  const int wl = (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int t_lo = (int)((long long)a.ntiles * wl / G);
  const int t_hi = (int)((long long)a.ntiles * (wl + 1) / G);
  if (t_lo >= t_hi) return;

  const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ws =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, UT * UNIT * 4, 0x00020000);

  // weight fragments: row = 32 j + l31 of the unit, 16-B chunk (2 g + h) ^ ((row >> 2) & 3): k = 8 g + 4 h + e in element e
  const int swz = (l31 >> 2) & 3;
  const int woff0 = l31 * 16 + (((0 + h) ^ swz) << 2);
  const int woff1 = l31 * 16 + (((2 + h) ^ swz) << 2);
  auto read_w = [&](f32x4 (&F)[NT], int slot, int g) {
    const float* base = ring + slot * UNIT + (g ? woff1 : woff0);
#pragma unroll
    for (int j = 0; j < NT; ++j) F[j] = *reinterpret_cast<const f32x4*>(base + j * 32 * 16);
  };

  // unit `unit` (of the stream, already wrapped) -> ring slot `slot`: a linear copy in 1-KB pieces (one wave instruction,
  // 16 B per lane), piece i by wave i % 8
  const unsigned dma_voff = (unsigned)lane * 16u;
  auto dma = [&](int unit, int slot) {
#pragma unroll
    for (int i = 0; i < (UNIT * 4) / 1024; i += 8) {
      if (i + wave < (UNIT * 4) / 1024) {
        float* dst = ring + slot * UNIT + (i + wave) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)dst, 16, (int)dma_voff, unit * (UNIT * 4) + (i + wave) * 1024, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // the barrier of a step: everything this wave has in flight has landed (VM = stores the wave may leave outstanding), then
  // the workgroup barrier publishes the unit the step's DMA brought (it is consumed two steps later, prefetched from one later)
  auto sync = [&](auto VM) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(decltype(VM)::value) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using VM0 = std::integral_constant<int, 0>;
  using VM4 = std::integral_constant<int, 4>;

  // ---- once per workgroup: biases into LDS, units 0 and 1 into the ring
  for (int i = tid; i < K::BIAS; i += 512) lbias[i] = a.bias[i];
  dma(0, 0);
  dma(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 Fw0[NT], Fw1[NT], FwN[NT];
  read_w(Fw0, 0, 0);

  // Tiles are walked DOWN the image (column-major inside an image), and the patch is a ring of PH rows: image row r lives in
  // patch row (r + 1) % PH.  A tile right below the one this workgroup has just finished finds its two top halo rows in
  // place and computes only its 8 new rows (8 x 34 = 272 patch pixels instead of 340: 9 fragment tiles instead of 11) --
  // "incremental"; the first tile of a workgroup and the top tile of every column compute all ten -- "full".
  // Head work of this wave, in fragment tiles of 32 flat patch pixels f = 32 t + l31 -> (f / PW, f % PW) of the rows computed:
  //   full:        tile `wave` (all three channel tiles) and, waves 0 .. NPT - 9, tile `wave + 8` (all three)
  //   incremental: tile `wave` (all three) and, waves 0 .. NT - 1, channel tile `wave` of tile 8 (the 16 leftover pixels)
  const bool two = wave < K::NPT - 8;
  static_assert(K::NPT - 8 == NT, "the same waves take the second tile (full) and one channel tile of tile 8 (incremental)");
  constexpr int PH = K::PH, NEW = K::TH * PW;
  int hf[3], hfy[3], hfx[3];                  // [0] tile wave, [1] tile wave + 8 (full), [2] tile 8 (incremental)
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    hf[p] = 32 * (p == 0 ? wave : p == 1 ? wave + 8 : 8) + l31;
    hfy[p] = hf[p] / PW;
    hfx[p] = hf[p] - hfy[p] * PW;
  }
  unsigned hv[2];                              // [0] tile wave, [1] the second tile of this wave in the mode of the tile
  bool hvalid[2], hin[2];
  int hpp[2];                                  // patch pixel the lane's result goes to
  bool h_inc = false;
  f32x4 X[3][2][2];                            // [K stage % 3][patch tile][k-group]: the pixel fragments travel two stages ahead
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
  // the head's pixel offsets of tile `tile` and its first K stage on the way (issued a step before the tile begins)
  auto head_setup = [&](int tile) {
    int tn, ty0, tx0;
    coords(tile, &tn, &ty0, &tx0, &h_inc);
    const int r0 = h_inc ? 2 : 0;
    const int ybase = ty0 % PH + r0;             // patch row of the first row computed, before the wrap
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
      hv[p] = hvalid[p] ? ((unsigned)((tn * a.H + iy) * a.W + ix) * (unsigned)(C * 4) + (unsigned)h * 16u) : kOOB;
      hpp[p] = pr * PW + fx;
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      X[st][0][0] = buf_load(xs, hv[0], st * 64);
      X[st][0][1] = buf_load(xs, hv[0], st * 64 + 32);
      if (two) {
        X[st][1][0] = buf_load(xs, hv[1], st * 64);
        X[st][1][1] = buf_load(xs, hv[1], st * 64 + 32);
      }
    }
  };
  head_setup(t_lo);

#ifdef SNTC_DIAG
  // make DIAG=1: cycles per phase (s_memtime around head / 3x3 / tail, summed over the workgroup's tiles) overwrite the first
  // floats of y at the end of the launch -- tools/rb_phases.py reads them; results are meaningless in such a build
  unsigned long long tph[4] = {0, 0, 0, 0};
#define RB_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define RB_STAMP(v)
#endif
  for (int tile = t_lo; tile < t_hi; ++tile) {
    bool inc_now;
    coords(tile, &n, &y0, &x0, &inc_now);
    RB_STAMP(ts0);

    // ================================================================================================
    // head: t1 = relu(W0 x + b0) on the halo patch, zero outside the image
    // ================================================================================================
    auto head = [&](auto NPXc, auto EXc) {
      constexpr int NPX = decltype(NPXc)::value;          // fragment tiles with all NT channel tiles
      constexpr bool EX = decltype(EXc)::value;           // + channel tile `wave` of one more fragment tile (its data in slot 1)
      constexpr int NLD = NPX + (EX ? 1 : 0);             // fragment tiles whose pixels this wave loads
      f32x16 acc[NT][NPX], accx;
#pragma unroll
      for (int e = 0; e < 16; ++e) accx[e] = 0.0f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < NPX; ++p)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[j][p][e] = 0.0f;
      const int wxoff = wave * 32 * 16;                   // EX: the unit's row block of channel tile `wave`
      static_for<0, U0>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr bool last = j == U0 - 1;
        // the pixel fragments of stage j + AHEAD are issued in step j (stages 0 and 1 came with head_setup): one step (1.5 us) is
        // not always enough for a first touch of x in HBM, two are; the two-tile variant (the first tile of a column, 1 in 32)
        // has no registers for a third buffer and stays one stage ahead
        constexpr int AHEAD = NPX == 2 ? 1 : 2;
        constexpr bool ld = j + AHEAD < U0 && j + AHEAD >= 2;
        dma(j + 2, (j + 2) % RING);
        if constexpr (ld) {
#pragma unroll
          for (int p = 0; p < NLD; ++p) {
            X[(j + AHEAD) % 3][p][0] = buf_load(xs, hv[p], (j + AHEAD) * 64);
            X[(j + AHEAD) % 3][p][1] = buf_load(xs, hv[p], (j + AHEAD) * 64 + 32);
          }
        }
        read_w(Fw1, j % RING, 1);
        f32x4 Fx0, Fx1;
        if constexpr (EX) {
          Fx0 = *reinterpret_cast<const f32x4*>(ring + (j % RING) * UNIT + wxoff + woff0);
          Fx1 = *reinterpret_cast<const f32x4*>(ring + (j % RING) * UNIT + wxoff + woff1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int p = 0; p < NPX; ++p) acc[jt][p] = RB_MFMA(Fw0[jt][e], X[j % 3][p][0][e], acc[jt][p]);
        if constexpr (EX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) accx = RB_MFMA(Fx0[e], X[j % 3][NPX][0][e], accx);
        }
        // every memory instruction behind an MFMA that covers its issue slot (mask 0x8 MFMA, 0x20 VMEM read, 0x100 DS read)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (ld) __builtin_amdgcn_sched_group_barrier(0x020, 2 * NLD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NT + (EX ? 2 : 0), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT * NPX + (EX ? 4 : 0) - 2, 0);
        read_w(FwN, (j + 1) % RING, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int p = 0; p < NPX; ++p) acc[jt][p] = RB_MFMA(Fw1[jt][e], X[j % 3][p][1][e], acc[jt][p]);
        if constexpr (EX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) accx = RB_MFMA(Fx1[e], X[j % 3][NPX][1][e], accx);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT * NPX + (EX ? 4 : 0) - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) Fw0[jt] = FwN[jt];
        if constexpr (last) {
          // registers 4 q .. 4 q + 3 of tile jt = channels 32 jt + 8 q + 4 h + (0..3) of pixel l31: chunk 2 (q & 1) + h of slab 2 jt + (q >> 1)
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
        if constexpr (ld && AHEAD == 2) sync(std::integral_constant<int, 2 * NLD>{});   // stage j + 2's fragments stay in flight across the barrier
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
    // 3x3: wave w = tile row w, lane = pixel, registers = the c/2 output channels
    // ================================================================================================
    RB_STAMP(ts1);
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    int bpp[3];                                // tap row dy of tile row `wave`: image row y0 - 1 + wave + dy, patch row (y0 + wave + dy) % PH
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) bpp[dy] = ((y0 + wave + dy) % PH) * PW + l31;
    auto px_off = [&](int tap, int g) {
      const int p = bpp[tap / 3] + (tap % 3);
      return p * 16 + ((((2 * g + h) ^ ((p >> 2) & 3))) << 2);
    };
    f32x4 Fp0 = *reinterpret_cast<const f32x4*>(patch + px_off(0, 0));
    f32x4 Fp1, FpN;
#pragma unroll 1
    for (int cc = 0; cc < SL; ++cc) {
      const float* pslab = patch + cc * (PP * 16);
      static_for<0, 9>([&](auto T) {
        constexpr int t = decltype(T)::value;
        dma(U0 + cc * 9 + t + 2, (t + 2) % RING);          // < UT: the last units belong to the tail
        read_w(Fw1, t % RING, 1);
        Fp1 = *reinterpret_cast<const f32x4*>(pslab + px_off(t, 1));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[jt] = RB_MFMA(Fw0[jt][e], Fp0[e], acc[jt]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 1, 0);
        read_w(FwN, (t + 1) % RING, 0);
        if constexpr (t < 8) FpN = *reinterpret_cast<const f32x4*>(pslab + px_off(t + 1, 0));
        else FpN = *reinterpret_cast<const f32x4*>(patch + min(cc + 1, SL - 1) * (PP * 16) + px_off(0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[jt] = RB_MFMA(Fw1[jt][e], Fp1[e], acc[jt]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) Fw0[jt] = FwN[jt];
        Fp0 = FpN;
        sync(VM0{});
      });
    }

    // ================================================================================================
    // tail: y = x + W2 relu(acc + b1) + b2; the accumulators are the B operand as they stand
    // ================================================================================================
    RB_STAMP(ts2);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + CH + 32 * jt + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jt][4 * q + e] = fmaxf(acc[jt][4 * q + e] + bv[e], 0.0f);
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
        dma((u + 2) % UT, (u + 2) % RING);                  // wraps into the next tile's head (harmless after the last tile)
        if constexpr (half == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) R[q] = buf_load(xs, poff[q], ot * 128);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
        }
        if constexpr (u == UT - 1) {
          if (tile + 1 < t_hi) head_setup(tile + 1);         // the next tile's first K stage travels under this step
          __builtin_amdgcn_sched_barrier(0);
        }
        read_w(Fw1, u % RING, 1);
        read_w(FwN, (u + 1) % RING, 0);
        // K = c/2 in stage order: stage st = NT * half + s of this unit, k-groups g = 0, 1, element e
        static_for<0, 2 * NT>([&](auto SG) {
          constexpr int sg = decltype(SG)::value, s = sg >> 1, g = sg & 1, st = NT * half + s;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2 = RB_MFMA((g ? Fw1 : Fw0)[s][e], acc[st >> 1][8 * (st & 1) + 4 * g + e], acc2);
          if constexpr (sg == 0) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
            if constexpr (half == 0) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) Fw0[jt] = FwN[jt];
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
          sync(VM4{});
        } else {
          sync(VM0{});
        }
      });
    });
#ifdef SNTC_DIAG
    RB_STAMP(ts3);
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

// wpack[u][row][16] in the ring's LDS image order, from the Keras kernels: w0 [1,1,C,CH], w1 [3,3,CH,CH], w2 [1,1,CH,C]
template <int C>
__global__ void __launch_bounds__(256) rb_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, float* __restrict__ wpack) {
  using K = RBCfg<C>;
  constexpr int CH = K::CH, NT = K::NT;
  const int total = K::UT * K::UNIT;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int u = idx / K::UNIT;
    const int rem = idx - u * K::UNIT;
    const int row = rem >> 4, pos = rem & 15;
    const int chunk = (pos >> 2) ^ ((row >> 2) & 3);
    const int k16 = chunk * 4 + (pos & 3);
    float v;
    if (u < K::U0) {
      v = w0[(size_t)(16 * u + k16) * CH + row];
    } else if (u < K::U0 + K::U1) {
      const int s = u - K::U0, cc = s / 9, t = s - cc * 9;
      v = w1[((size_t)t * CH + 16 * cc + k16) * CH + row];
    } else {
      const int s = u - K::U0 - K::U1, ot = s >> 1, half = s & 1;
      const int st = NT * half + (row >> 5);
      v = w2[(size_t)(16 * st + k16) * C + 32 * ot + (row & 31)];
    }
    wpack[idx] = v;
  }
}

__global__ void rb_bias_kernel(const float* b0, const float* b1, const float* b2, float* out, int ch, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ch) out[i] = b0 ? b0[i] : 0.0f;
  else if (i < 2 * ch) out[i] = b1 ? b1[i - ch] : 0.0f;
  else if (i < 2 * ch + c) out[i] = b2 ? b2[i - 2 * ch] : 0.0f;
}

constexpr int kMaxDev = 16;
struct RBDevice {
  std::once_flag once;
  int rc = SNTC_OK;
  int num_cus = 0;
};
RBDevice g_rbdev[kMaxDev];

int rb_init(int* num_cus) {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the ResidualBlock tables");
  RBDevice& D = g_rbdev[dev];
  std::call_once(D.once, [&] {
    D.rc = [&]() -> int {
      hipDeviceProp_t prop;
      SNTC_HIP(hipGetDeviceProperties(&prop, dev));
      D.num_cus = prop.multiProcessorCount;
      SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)RBCfg<192>::LDS));
      return SNTC_OK;
    }();
  });
  *num_cus = D.num_cus;
  return D.rc;
}

}  // namespace
}  // namespace sntc

struct sntc_resblock_plan {
  int c = 0;
  int precision = 0;          // 0: exact fp32 (rb_fused.hip); 1: bf16 x 3 split precision (rb_fused_bf3.hip)
  float* wpack = nullptr;
  void* wpack3 = nullptr;     // the pre-split weight stream (precision 1 only)
  float* bias = nullptr;
  int max_workgroups = 0;     // 0: one per CU
};

using namespace sntc;

static int rb_pack(sntc_resblock_plan* p, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                   const float* b2, hipStream_t s) {
  using K = RBCfg<192>;
  hipLaunchKernelGGL(rb_pack_kernel<192>, dim3((K::UT * K::UNIT + 255) / 256), dim3(256), 0, s, w0, w1, w2, p->wpack);
  hipLaunchKernelGGL(rb_bias_kernel, dim3((K::BIAS + 255) / 256), dim3(256), 0, s, b0, b1, b2, p->bias, K::CH, 192);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ResidualBlock weight packing");
  if (p->wpack3) return rb3_pack(w0, w1, w2, p->wpack3, s);
  return SNTC_OK;
}

extern "C" int sntc_resblock_supported(int c) { return c == 192 ? 1 : 0; }

static void rb_free(sntc_resblock_plan* p) {
  if (p->wpack) (void)hipFree(p->wpack);
  if (p->wpack3) (void)hipFree(p->wpack3);
  if (p->bias) (void)hipFree(p->bias);
  delete p;
}

extern "C" int sntc_resblock_plan_create(int c, const float* w0, const float* b0, const float* w1, const float* b1,
                                         const float* w2, const float* b2, int precision, void* stream, sntc_resblock_plan** plan) {
  if (!plan || !w0 || !w1 || !w2) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_plan_create: null argument");
  if (!sntc_resblock_supported(c)) return fail(SNTC_ERR_UNSUPPORTED, "sntc_resblock_plan_create: fused ResidualBlock exists for c = 192");
  if (precision != 0 && precision != 1) return fail(SNTC_ERR_UNSUPPORTED, "sntc_resblock_plan_create: precision 0 (fp32) or 1 (bf16 x 3)");
  int cus = 0;
  if (int rc = rb_init(&cus)) return rc;
  if (precision == 1)
    if (int rc = rb3_init()) return rc;
  using K = RBCfg<192>;
  auto* p = new sntc_resblock_plan();
  p->c = c;
  p->precision = precision;
  if (hipMalloc(&p->wpack, sizeof(float) * K::UT * K::UNIT) != hipSuccess || hipMalloc(&p->bias, sizeof(float) * K::BIAS) != hipSuccess ||
      (precision == 1 && hipMalloc(&p->wpack3, rb3_pack_bytes()) != hipSuccess)) {
    rb_free(p);
    return fail(SNTC_ERR_HIP, "sntc_resblock_plan_create: out of device memory");
  }
  if (int rc = rb_pack(p, w0, b0, w1, b1, w2, b2, (hipStream_t)stream)) {
    rb_free(p);
    return rc;
  }
  *plan = p;
  return SNTC_OK;
}

extern "C" int sntc_resblock_plan_update(sntc_resblock_plan* p, const float* w0, const float* b0, const float* w1, const float* b1,
                                         const float* w2, const float* b2, void* stream) {
  if (!p || !w0 || !w1 || !w2) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_plan_update: null argument");
  return rb_pack(p, w0, b0, w1, b1, w2, b2, (hipStream_t)stream);
}

extern "C" void sntc_resblock_plan_destroy(sntc_resblock_plan* p) {
  if (p) rb_free(p);
}

extern "C" int sntc_resblock_plan_set_workgroups(sntc_resblock_plan* p, int max_workgroups) {
  if (!p || max_workgroups < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_plan_set_workgroups: bad argument");
  p->max_workgroups = max_workgroups;
  return SNTC_OK;
}

extern "C" int64_t sntc_resblock_flops(const sntc_resblock_plan* p, int n, int h, int w) {
  if (!p || n < 0 || h < 0 || w < 0) return -1;
  const int64_t c = p->c, ch = p->c / 2;
  return 2 * (int64_t)n * h * w * (c * ch + 9 * ch * ch + ch * c);
}

extern "C" int sntc_resblock_forward(const sntc_resblock_plan* p, const float* x, int n, int h, int w, float* y, void* stream) {
  if (!p || !x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_forward: null argument");
  if (n < 0 || h < 0 || w < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_forward: negative size");
  if (x == y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_forward: y must not alias x (tiles read their neighbours' halo)");
  if (n == 0 || h == 0 || w == 0) return SNTC_OK;
  using K = RBCfg<192>;
  const int64_t bytes = (int64_t)n * h * w * p->c * 4;
  if (bytes >= (1LL << 31)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_forward: tensor of 2 GiB or more; split the batch");
  int cus = 0;
  if (int rc = rb_init(&cus)) return rc;
  RBArgs a{};
  a.x = x; a.y = y; a.wpack = p->wpack; a.bias = p->bias;
  a.bytes = (unsigned)bytes;
  a.N = n; a.H = h; a.W = w;
  a.tiles_x = (w + K::TW - 1) / K::TW;
  a.tiles_y = (h + K::TH - 1) / K::TH;
  const int64_t nt = (int64_t)n * a.tiles_x * a.tiles_y;
  if (nt >= (1LL << 31)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_resblock_forward: too many tiles");
  a.ntiles = (int)nt;
  int grid = std::min<int64_t>(nt, p->max_workgroups > 0 ? p->max_workgroups : cus);
  if (p->precision == 1) {
    a.wpack = reinterpret_cast<const float*>(p->wpack3);
    return rb3_launch(a, grid, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(rb_kernel<192>, dim3(grid), dim3(512), K::LDS, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ResidualBlock launch");
  return SNTC_OK;
}

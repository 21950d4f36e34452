// syn_fused.hip -- the first layer of the two-layer syntheses (reference common/transforms.py:298-361) as ONE launch:
//
//     hidden = act(base_conv(y_hat)) [+ res(y_hat)]            TwoLayerSynthesis / TwoLayerResSynthesis, :315 / :355-359
//
// base_conv (and the convolutional residual branch, whose kernel is concatenated along Cout: both read y_hat) is a
// Conv2DTranspose k x k / s, SAME (13 x 13 / 8 in every reference config); act is IGDN1 / GDN1 / relu / leaky-relu / none.
// The output is the HIDDEN tensor [n, s h, s w, ch] -- what the reference hands to its output convolution -- not the
// [base | res] pair the generic path writes: the activation and the residual add happen on the accumulators.
//
// Frame.  Output pixel (Y, X) = (s Q + r), r in [0, s): with pt = (k - s) / 2, phi = (r + pt) % s and qo = (r + pt) / s the
// taps of residue r are j = 0 .. cnt(r) - 1, kernel index phi + s j, SOURCE latent pixel Q + qo - j.  Indexed by the output-
// aligned macro pixel Q the sources are Q + d with d in {+1, 0, -1} (13 / 8: r = 0..2 -> d in {0, -1}; r = 3..5 -> {0}; r = 6, 7
// -> {+1, 0}), so the macro grid IS the latent grid (no (h + 1) x (w + 1) grid with half-valid border rows as in the
// phase-grouped gather GEMM) and a tap is one of nine whole-pixel SHIFTS (dy, dx) of the input, si = 3 (1 - dy) + (1 - dx).
//
// Work.  An item = (tile of 256 consecutive latent pixels of one image, UNIT of 96 output columns).  A unit is 96 / cp
// output phases (cp = columns per phase: 24 for 12 + 12 residual channels) that share their shift set (the leftovers of
// the nine phase classes are grouped so that shift sets nest; a column whose phase does not use a step's shift carries a
// zero weight there).  The workgroup (512 threads = 8 waves, one per CU) walks the unit's K loop -- channel slab outermost,
// the unit's shifts inside, 16 channels per step -- with
//   * the weights as MFMA A operand: one 6-KB ring unit per step (96 rows x 16 k, the LDS image as packed), LDS-DMA into an
//     eight-slot ring, five units ahead of the step that reads them (a counted vmcnt leaves the youngest in flight);
//   * the pixels as B operand: wave w owns pixels 32 w .. 32 w + 31 of the tile; the 16-channel slab of the tile's PATCH
//     (the tile's 256 pixels + one image row + one pixel either side, flat) is staged once per slab by LDS-DMA (3 slots) and
//     every shift is a shifted fragment read of it ([pixel][16] with the 16-B chunks XOR-swizzled by (pixel >> 2) & 3:
//     conflict-free under any shift); a source outside the image reads the slot's zero row instead (one select on the address);
//   * 24 MFMAs per wave and step (16 where the unit's third tile is empty for the step's shift), fragments double-buffered
//     under them; the loop is compiled per shift count (1 / 2 / 4 per slab) with one barrier per PAIR of steps, and in a
//     generic form (run-time shift, one barrier per step) for everything else.
// Items are dealt dynamically (one atomic per item, fetched two items ahead), most expensive units first, unit-major so that
// the workgroups of an XCD stream the same weights at about the same time; the next item's first patch slabs and ring
// units are in flight while the current item's epilogue runs.  Per-image geometry comes with the item: batches of
// different shapes (the Kodak set's two orientations) share one launch.
//
// Same bits as the generic path (gather_gemm.hip phase groups + pixel.hip's tail, stage 1): every output is the same
// k-ordered fp32 fma chain -- slab outermost, taps (jy, jx) ascending = shifts si ascending, k in {8g+e, 8g+4+e} per MFMA;
// products commute; a zero weight or a zero (padding) pixel contributes fma(x, 0, acc) = acc -- then + bias, the activation
// with the norm pool summed in channel order with separately rounded products (this file is compiled -ffp-contract=off like
// pixel.hip), then + residual.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "sntc_internal.h"

#include "rb_common.h"

namespace sntc {
namespace syn {

using rb::buf_load;
using rb::buf_store;
using rb::kOOB;

constexpr int kNT = 3;                      // 32-row MFMA tiles per unit
constexpr int kRows = 32 * kNT;             // weight rows (output columns) per unit
constexpr int kUnit = kRows * 16;           // floats per ring unit (one 16-deep step): 6144 B
constexpr int kRing = 8;                     // ring units: the DMA runs kLead units ahead of the step that reads them
constexpr int kLead = 5;
constexpr int kPSlots = 3;
constexpr int kPPMax = 512;                 // patch pixels per slot: 256 + 2 w + 2 <= 512 -> w <= 127
constexpr int kPStride = (kPPMax + 1) * 16; // floats per patch slot: the pixels, then one row of zeros (what a source outside the image reads)
constexpr int kTileM = 256;
constexpr int kMaxSynGroups = 4;
constexpr int kMaxSlots = 8;

struct SynUnit {              // 64 B, read with scalar loads
  int ns;                     // steps per 16-channel slab = shifts of this unit
  unsigned sl0, sl1;          // shift index of step j: 4 bits each (sl0: j = 0..7, sl1: j = 8)
  int step0;                  // first ring unit of this unit in the packed stream
  unsigned ph[kMaxSlots];     // phase of slot q: (ry << 8) | rx, 0xffffffff = empty
  int cost;                   // 32-row tile steps per slab
  unsigned pm;                // bit j: step j is PARTIAL -- the unit's third tile holds no phase that uses its shift
  int pad[2];
};
static_assert(sizeof(SynUnit) == 64, "SynUnit is read as 16 dwords");

struct SynGeom {              // the layer
  int k, s, pt, cin, cp;      // kernel, stride, pad before, input channels, columns per phase
};

__host__ __device__ inline int unit_shift(const SynUnit& u, int j) { return (int)(((j < 8 ? u.sl0 >> (4 * j) : u.sl1 >> (4 * (j - 8)))) & 15u); }

// weight of ring-unit row R, stage position k16 of (unit u, slab cc, step j): w is the Keras transposed kernel [k][k][cp][cin]
__host__ __device__ inline float pack_value(const SynGeom& G, const SynUnit& u, int cc, int j, int R, int k16, const float* w) {
  const int jt = R >> 5, i = R & 31, h = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
  const int vi = 16 * jt + r;                         // value index inside the lane half: 0 .. 47
  const int sph = 48 / G.cp;
  const unsigned ph = u.ph[h * sph + vi / G.cp];
  if (ph == 0xffffffffu) return 0.0f;
  const int ch = vi % G.cp;
  const int ry = (int)(ph >> 8), rx = (int)(ph & 255u);
  const int si = unit_shift(u, j), dy = 1 - si / 3, dx = 1 - si % 3;
  const int phy = (ry + G.pt) % G.s, qoy = (ry + G.pt) / G.s, cy = (G.k - phy + G.s - 1) / G.s;
  const int phx = (rx + G.pt) % G.s, qox = (rx + G.pt) / G.s, cx = (G.k - phx + G.s - 1) / G.s;
  const int jy = qoy - dy, jx = qox - dx;
  if (jy < 0 || jy >= cy || jx < 0 || jx >= cx) return 0.0f;
  const int ky = phy + G.s * jy, kx = phx + G.s * jx;
  return w[((size_t)(ky * G.k + kx) * G.cp + ch) * G.cin + cc * 16 + k16];
}

struct SynGroup {
  const float* x;   // [n, h, w, cin]
  float* v;         // [n, s h, s w, ch]
  int n, h, w;
  int tile0;        // first tile of this group
  int tpi;          // tiles per image
};

struct SynArgs {
  const float* wpack;
  const SynUnit* units;      // sorted: most expensive first
  const float* tables;       // bias[cp] | beta[ch] | gamma[ch * ch]
  int* queue;                // item counter, zero at launch
  unsigned wbytes;
  int nunits, ntiles, nitems;
  int cin, nslab, s, act_kind;
  int ngroups;
  int dbg;                   // -DSNTC_DIAG builds only (SNTC_SYN_DBG): 1 no epilogue, 2 no patch DMA, 4 no ring DMA, 8 no MFMAs, 16 no
                             // barriers -- what a launch waits on; results are meaningless with any of these bits set; 32: the generic
                             // K loop for every unit (correct results)
  SynGroup g[kMaxSynGroups];
};

#define SYN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The kernel arguments through an OPAQUE pointer to the kernarg segment (gather_gemm.hip): what runs once per item (item
// decoding, loader set-up, epilogue) re-reads its scalars from the scalar cache instead of pinning the whole argument block
// (four batch records) in SGPRs through the K loop.
typedef const SynArgs __attribute__((address_space(4))) SynKArgs;
__device__ __forceinline__ SynKArgs& syn_args() {
  SynKArgs* kp = (SynKArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *kp;
}

template <int CP, bool RES>
struct SynCfg {
  static constexpr int CH = RES ? CP / 2 : CP;
  static constexpr int SPH = 48 / CP;                  // phase slots per lane half
  static constexpr int TAB = CP + CH + CH * CH;        // floats: bias | beta | gamma
  static constexpr size_t LDS = (size_t)(kPSlots * kPStride + kRing * kUnit + TAB + 16) * 4;
  static_assert(48 % CP == 0, "a lane half holds whole phases");
};

template <int CP, bool RES>
__global__ void __launch_bounds__(512, 2) syn_kernel(const SynArgs a) {
  using K = SynCfg<CP, RES>;
  constexpr int CH = K::CH, SPH = K::SPH;
  typedef __attribute__((address_space(3))) void lds_void;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);           // [kPSlots][kPPMax pixels + one row of zeros][16]
  float* ring = patch + kPSlots * kPStride;                // [kRing][96][16]
  float* tab = ring + kRing * kUnit;                       // bias | beta | gamma
  int* sh_item = reinterpret_cast<int*>(tab + K::TAB);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int i = tid; i < K::TAB; i += 512) tab[i] = a.tables[i];
  if (tid < 16 * kPSlots) patch[(tid >> 4) * kPStride + kPPMax * 16 + (tid & 15)] = 0.0f;

  const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, (int)a.wbytes, 0x00020000);

  // weight fragments (rb_fused.hip): row = 32 jt + l31 of the ring unit, 16-B chunk (2 g + h) ^ ((row >> 2) & 3)
  const int swz = (l31 >> 2) & 3;
  const int woff0 = l31 * 16 + (((0 + h) ^ swz) << 2);
  const int woff1 = l31 * 16 + (((2 + h) ^ swz) << 2);
  auto read_w = [&](f32x4 (&F)[kNT], int slot, int g) {
    const float* base = ring + slot * kUnit + (g ? woff1 : woff0);
#pragma unroll
    for (int j = 0; j < kNT; ++j) F[j] = *reinterpret_cast<const f32x4*>(base + j * 32 * 16);
  };
  const unsigned dma_voff = (unsigned)lane * 16u;
  auto dma_w = [&](int gstep, int slot) {            // one ring unit = six 1-KB pieces, piece i by wave i
    if (wave < (kUnit * 4) / 1024) {
      float* dst = ring + slot * kUnit + wave * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)dst, 16, (int)dma_voff, gstep * (kUnit * 4) + wave * 1024, 0, 0);
    }
  };

  // ---- item state
  struct Item {
    int u, gi, n, m0, H, W;
  };
  auto decode = [&](int item, Item* it) {
    SynKArgs& a = syn_args();
    const int u = item / a.ntiles;
    const int t = item - u * a.ntiles;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < kMaxSynGroups; ++i)
      if (i < a.ngroups && t >= a.g[i].tile0) gi = i;
    const int tl = t - a.g[gi].tile0;
    const int n = tl / a.g[gi].tpi;
    it->u = u; it->gi = gi; it->n = n; it->m0 = (tl - n * a.g[gi].tpi) * kTileM;
    it->H = a.g[gi].h; it->W = a.g[gi].w;
  };

  // per-item loader state: the patch pieces of this wave (piece wave + 8 i: 16 patch pixels, 4 lanes x 16 B each)
  unsigned pvoff[4];
  int npieces = 0;
  __amdgpu_buffer_rsrc_t xs = ws;
  auto setup_loader = [&](const Item& it) {
    SynKArgs& a = syn_args();
    const int HW = it.H * it.W;
    const int p0 = it.m0 - it.W - 1;
    npieces = (kTileM + 2 * it.W + 2 + 15) >> 4;
    xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[it.gi].x) + (size_t)it.n * HW * a.cin, 0, HW * a.cin * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = 16 * (wave + 8 * i) + (lane >> 2);
      const int msrc = p0 + pp;
      const int c = (lane & 3) ^ ((pp >> 2) & 3);
      pvoff[i] = (unsigned)msrc < (unsigned)HW ? (unsigned)msrc * (unsigned)(a.cin * 4) + (unsigned)c * 16u : kOOB;
    }
  };
  auto dma_patch = [&](int cc, int pslot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wave + 8 * i < npieces) {
        float* dst = patch + pslot * kPStride + (wave + 8 * i) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)dst, 16, (int)pvoff[i], cc * 64, 0, 0);
      }
  };

  __syncthreads();
  int item = blockIdx.x;
  if (item >= a.nitems) return;
  Item I{};
  decode(item, &I);
  setup_loader(I);
  struct UnitHot {
    int ns, step0;
    unsigned sl0, sl1, pm;
  };
  auto load_unit = [&](int u, UnitHot* U) {
    const SynUnit* up = syn_args().units + u;
    U->ns = up->ns; U->step0 = up->step0; U->sl0 = up->sl0; U->sl1 = up->sl1; U->pm = up->pm;
  };
  auto shift_of = [&](const UnitHot& U, int j) { return (int)(((j < 8 ? U.sl0 >> (4 * j) : U.sl1 >> (4 * (j - 8)))) & 15u); };
  const int nslab = a.nslab;
  UnitHot U;
  load_unit(I.u, &U);
  // an item's first patch slabs and ring units (the ring runs kLead units ahead of the step that reads them)
  auto prologue = [&](const UnitHot& U) {
    const int T = nslab * U.ns;
    dma_patch(0, 0);
    if (nslab > 1) dma_patch(1, 1);
#pragma unroll
    for (int u = 0; u < kLead; ++u)
      if (u < T) dma_w(U.step0 + u, u);
  };
  prologue(U);
  // the work queue, two items ahead: the atomic is issued behind an item's prologue, returns under the previous item's
  // epilogue and is drained by the wait that opens the item -- no step of the K loop ever waits for it
  int fetched = 0;
  if (tid == 0) fetched = atomicAdd(syn_args().queue, 1) + (int)gridDim.x;
  int islot = 0;        // sh_item is double-buffered by item parity: slot p is rewritten two items later, i.e. behind a barrier that
                        // every wave reaches only after its read of slot p (a single slot was a formal LDS race: ADVICE r5)

  while (true) {
    // ---- this item's lane geometry
    const int HW = I.H * I.W;
    const int m = I.m0 + 32 * wave + l31;
    const bool pvalid = m < HW;
    const int qy = m / I.W, qx = m - qy * I.W;
    unsigned vmask = 0;
#pragma unroll
    for (int si = 0; si < 9; ++si) {
      const int dy = 1 - si / 3, dx = 1 - si % 3;
      if (pvalid && (unsigned)(qy + dy) < (unsigned)I.H && (unsigned)(qx + dx) < (unsigned)I.W) vmask |= 1u << si;
    }
    const int lanepp = 32 * wave + l31 + I.W + 1;
    const int zoff = kPPMax * 16;              // the slot's row of zeros, relative to the slot
    // patch fragment: pixel lanepp + dy W + dx of slab slot `ps`, chunk (2 g + h) ^ ((pixel >> 2) & 3); the zero row if invalid
    auto read_p = [&](int ps, int si, int g) -> f32x4 {
      const int q3 = si / 3;
      const int pp = lanepp + (1 - q3) * I.W + 1 - (si - 3 * q3);             // + dy W + dx
      const int off = pp * 16 + ((((2 * g + h) ^ ((pp >> 2) & 3))) << 2);
      const int sel = -(int)((vmask >> si) & 1u);                            // branch-free select (all ones: in the image)
      return *reinterpret_cast<const f32x4*>(patch + (ps * kPStride + ((off & sel) | (zoff & ~sel))));
    };

    const int ns = U.ns;
    const int T = nslab * ns;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (tid == 0) sh_item[islot] = fetched;    // the item after this one (read by everybody behind the K loop's barriers)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc[kNT];
#pragma unroll
    for (int jt = 0; jt < kNT; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jt][e] = 0.0f;

    f32x4 Fw0[kNT], Fw1[kNT], FwN[kNT];
    f32x4 Fp0, Fp1, FpN;
    int cc = 0, j = 0, ps = 0;                 // slab, step inside the slab, patch slot of the slab (cc % 3)
    int si = shift_of(U, 0);

    // One step = one ring unit (the generic form: any number of shifts per slab).  MASKED (the units whose phases do not all share one shift set): where the unit's third tile
    // holds only phases that do not use this step's shift its weights are zeros and its 8 MFMAs are left out (fma(x, 0, acc) = acc).
    auto step = [&](int t, auto MASKEDc) {
      constexpr bool MASKED = decltype(MASKEDc)::value;
      // coordinates of the next step
      int jn = j + 1, ccn = cc, psn = ps;
      if (jn == ns) {
        jn = 0;
        ccn = cc + 1;
        psn = ps == kPSlots - 1 ? 0 : ps + 1;
      }
      const int sin = shift_of(U, jn);
      const bool third = !MASKED || !((U.pm >> j) & 1u);
      // patch slab cc + 2 first, ring unit t + kLead last: the step's wait then leaves exactly that ring unit in flight
      if (j == 0 && cc + 2 < nslab && !SNTC_DBG(a, 2)) dma_patch(cc + 2, ps == 0 ? 2 : ps - 1);
      const bool wdma = t + kLead < T && !SNTC_DBG(a, 4);
      if (wdma) dma_w(U.step0 + t + kLead, (t + kLead) & (kRing - 1));
      __builtin_amdgcn_sched_barrier(0);
      read_w(Fw1, t & (kRing - 1), 1);
      Fp1 = read_p(ps, si, 1);
      if (!SNTC_DBG(a, 8)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) acc[jt] = SYN_MFMA(Fw0[jt][e], Fp0[e], acc[jt]);
      }
      if (!MASKED) {
        if (!SNTC_DBG(a, 8)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[kNT - 1] = SYN_MFMA(Fw0[kNT - 1][e], Fp0[e], acc[kNT - 1]);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kNT + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * kNT - 1, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kNT + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * (kNT - 1) - 1, 0);
      }
      read_w(FwN, (t + 1) & (kRing - 1), 0);
      FpN = read_p(psn, sin, 0);
      if (!SNTC_DBG(a, 8)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) acc[jt] = SYN_MFMA(Fw1[jt][e], Fp1[e], acc[jt]);
      }
      if (!MASKED) {
        if (!SNTC_DBG(a, 8)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[kNT - 1] = SYN_MFMA(Fw1[kNT - 1][e], Fp1[e], acc[kNT - 1]);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kNT + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * kNT - 1, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kNT + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * (kNT - 1) - 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MASKED && third && !SNTC_DBG(a, 8)) {            // the third tile's eight MFMAs of this step, k order as everywhere
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[kNT - 1] = SYN_MFMA(Fw0[kNT - 1][e], Fp0[e], acc[kNT - 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[kNT - 1] = SYN_MFMA(Fw1[kNT - 1][e], Fp1[e], acc[kNT - 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jt = 0; jt < kNT; ++jt) Fw0[jt] = FwN[jt];
      Fp0 = FpN;
      cc = ccn; j = jn; ps = psn; si = sin;
      // everything older than this step's ring unit has landed (vmcnt counts in issue order): ring unit t + kLead - 1 and the patch slab
      if (wdma && wave < (kUnit * 4) / 1024) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (!SNTC_DBG(a, 16)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // The same loop with the unit's shifts (1, 2 or 4 per slab) and partial-step mask known at compile time: one slab per
    // iteration, its steps unrolled -- the fragment addresses of the shifts are computed once per item (g = 1: the g = 0 address
    // with the chunk's bit 1 flipped), a step opens with an MFMA (the DMA issue sits behind it), and an even number of steps
    // shares one barrier per PAIR of steps (the ring is deep enough: unit t + kLead is issued in step t).
    auto kloop = [&](auto NSc, auto PMc) {
      constexpr int NS = decltype(NSc)::value;
      constexpr unsigned PM = decltype(PMc)::value;
      constexpr bool PAIR = NS % 2 == 0;
      int pa[NS];
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) {
        const int sj = shift_of(U, jj);
        const int q3 = sj / 3;
        const int pp = lanepp + (1 - q3) * I.W + 1 - (sj - 3 * q3);
        const int off = pp * 16 + (((h ^ ((pp >> 2) & 3))) << 2);
        const int sel = -(int)((vmask >> sj) & 1u);
        pa[jj] = (off & sel) | (zoff & ~sel);
      }
      read_w(Fw0, 0, 0);
      Fp0 = *reinterpret_cast<const f32x4*>(patch + pa[0]);
      int psl = 0;
#pragma unroll 1
      for (int cs = 0; cs < nslab; ++cs) {
        const int psoff = psl * kPStride;
        const int psn = psl == kPSlots - 1 ? 0 : psl + 1;
        const int psoffn = psn * kPStride;
        rb::static_for<0, NS>([&](auto Jc) {
          constexpr int jj = decltype(Jc)::value;
          constexpr bool third = !((PM >> jj) & 1u);
          constexpr bool third_next = !((PM >> ((jj + 1) % NS)) & 1u);
          constexpr int NTA = third ? kNT : kNT - 1;
          constexpr int NTN = third_next ? kNT : kNT - 1;
          const int t = cs * NS + jj;
          if (!SNTC_DBG(a, 8)) acc[0] = SYN_MFMA(Fw0[0][0], Fp0[0], acc[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (jj == 0 && cs + 2 < nslab && !SNTC_DBG(a, 2)) dma_patch(cs + 2, psl == 0 ? 2 : psl - 1);
          const bool wdma = t + kLead < T && !SNTC_DBG(a, 4);
          if (wdma) dma_w(U.step0 + t + kLead, (t + kLead) & (kRing - 1));
          __builtin_amdgcn_sched_barrier(0);
          {
            const float* base = ring + (t & (kRing - 1)) * kUnit + woff1;
#pragma unroll
            for (int jt = 0; jt < NTA; ++jt) Fw1[jt] = *reinterpret_cast<const f32x4*>(base + jt * 32 * 16);
          }
          Fp1 = *reinterpret_cast<const f32x4*>(patch + ((pa[jj] ^ 8) + psoff));
          if (!SNTC_DBG(a, 8)) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int jt = 0; jt < NTA; ++jt)
                if (e + jt > 0) acc[jt] = SYN_MFMA(Fw0[jt][e], Fp0[e], acc[jt]);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, NTA + 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTA - 1, 0);
          {
            const float* base = ring + ((t + 1) & (kRing - 1)) * kUnit + woff0;
#pragma unroll
            for (int jt = 0; jt < NTN; ++jt) FwN[jt] = *reinterpret_cast<const f32x4*>(base + jt * 32 * 16);
          }
          if constexpr (jj + 1 < NS) FpN = *reinterpret_cast<const f32x4*>(patch + (pa[jj + 1] + psoff));
          else FpN = *reinterpret_cast<const f32x4*>(patch + (pa[0] + psoffn));
          if (!SNTC_DBG(a, 8)) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int jt = 0; jt < NTA; ++jt) acc[jt] = SYN_MFMA(Fw1[jt][e], Fp1[e], acc[jt]);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NTN + 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTA - 1, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jt = 0; jt < kNT; ++jt) Fw0[jt] = FwN[jt];
          Fp0 = FpN;
          if constexpr (!PAIR) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (!SNTC_DBG(a, 16)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (jj & 1) {
            // leave this pair's ring units in flight (the youngest instructions of the wave): everything older has landed
            if (wave < (kUnit * 4) / 1024 && t - 1 + kLead < T && !SNTC_DBG(a, 4)) {
              if (t + kLead < T) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            if (!SNTC_DBG(a, 16)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        psl = psn;
      }
    };
    using N1 = std::integral_constant<int, 1>;
    using N2 = std::integral_constant<int, 2>;
    using N4 = std::integral_constant<int, 4>;
    const bool generic = SNTC_DBG(a, 32);
    if (!generic && ns == 4 && U.pm == 0) kloop(N4{}, std::integral_constant<unsigned, 0>{});
    else if (!generic && ns == 4 && U.pm == 0xcu) kloop(N4{}, std::integral_constant<unsigned, 0xc>{});
    else if (!generic && ns == 4 && U.pm == 0xau) kloop(N4{}, std::integral_constant<unsigned, 0xa>{});
    else if (!generic && ns == 2 && U.pm == 0) kloop(N2{}, std::integral_constant<unsigned, 0>{});
    else if (!generic && ns == 1 && U.pm == 0) kloop(N1{}, std::integral_constant<unsigned, 0>{});
    else if (U.pm == 0) {
      read_w(Fw0, 0, 0);
      Fp0 = read_p(0, si, 0);
#pragma unroll 1
      for (int t = 0; t < T; ++t) step(t, std::integral_constant<bool, false>{});
    } else {
      read_w(Fw0, 0, 0);
      Fp0 = read_p(0, si, 0);
#pragma unroll 1
      for (int t = 0; t < T; ++t) step(t, std::integral_constant<bool, true>{});
    }

    // ---- the next item: its first patch slabs and ring units travel while this item's results leave
    const int item_next = __builtin_amdgcn_readfirstlane(sh_item[islot]);
    islot ^= 1;
    const Item Icur = I;
    const bool more = item_next < syn_args().nitems;
    if (more) {
      decode(item_next, &I);
      setup_loader(I);
      load_unit(I.u, &U);
      prologue(U);
      if (tid == 0) fetched = atomicAdd(syn_args().queue, 1) + (int)gridDim.x;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane = pixel; the lane half holds SPH whole phases: value vi = 16 jt + r <-> (slot vi / CP, channel vi % CP)
    if (!SNTC_DBG(a, 1)) {
      SynKArgs& a = syn_args();
      const SynUnit* up = a.units + Icur.u;
      const int Wo = Icur.W * a.s;
      const __amdgpu_buffer_rsrc_t vs = __builtin_amdgcn_make_buffer_rsrc(
          a.g[Icur.gi].v + (size_t)Icur.n * HW * (a.s * a.s) * CH, 0, HW * (a.s * a.s) * CH * 4, 0x00020000);
      const float* bias = tab;
      const float* beta = tab + CP;
      const float* gamma = tab + CP + CH;
      // One phase at a time, the norm pool two output channels at a time: nrm[j], nrm[j + 1] += |t_i| * (gamma[i][j], gamma[i][j + 1])
      // -- the gamma pair is one LDS read (a broadcast: every lane reads the same address), the two-element multiply and add are
      // the same IEEE operations per element as the scalar ones (separately rounded: this file is built -ffp-contract=off), in
      // the order of pixel.hip's tail (i ascending for every j).  The epilogue is VALU-issue bound (two waves per SIMD are in it
      // at once), so instructions are what counts.
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const int act = a.act_kind;
#pragma unroll
      for (int sl = 0; sl < SPH; ++sl) {
        float tv[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const int vi = sl * CP + c;
          tv[c] = acc[vi >> 4][vi & 15] + bias[c];
        }
        float o[CH];
        if (act == 1 || act == 2) {
          float av[CH];
#pragma unroll
          for (int ic = 0; ic < CH; ++ic) av[ic] = fabsf(tv[ic]);
          f32x2 nrm[CH / 2];
#pragma unroll
          for (int jp = 0; jp < CH / 2; ++jp) nrm[jp] = *reinterpret_cast<const f32x2*>(beta + 2 * jp);
#pragma unroll
          for (int ic = 0; ic < CH; ++ic)
#pragma unroll
            for (int jp = 0; jp < CH / 2; ++jp) {
              const f32x2 g = *reinterpret_cast<const f32x2*>(gamma + ic * CH + 2 * jp);
              const f32x2 pr = g * av[ic];
              nrm[jp] = nrm[jp] + pr;
            }
          if (act == 1) {
#pragma unroll
            for (int jc = 0; jc < CH; ++jc) o[jc] = tv[jc] * nrm[jc >> 1][jc & 1];
          } else {
#pragma unroll
            for (int jc = 0; jc < CH; ++jc) o[jc] = tv[jc] / nrm[jc >> 1][jc & 1];
          }
        } else if (act == 3) {
#pragma unroll
          for (int jc = 0; jc < CH; ++jc) o[jc] = fmaxf(tv[jc], 0.0f);
        } else if (act == 4) {
#pragma unroll
          for (int jc = 0; jc < CH; ++jc) o[jc] = tv[jc] >= 0.0f ? tv[jc] : 0.2f * tv[jc];
        } else {
#pragma unroll
          for (int jc = 0; jc < CH; ++jc) o[jc] = tv[jc];
        }
        if (RES) {
#pragma unroll
          for (int jc = 0; jc < CH; ++jc) o[jc] = o[jc] + tv[CH + jc];
        }
        const unsigned ph0 = up->ph[sl], ph1 = up->ph[SPH + sl];
        const unsigned ph = h ? ph1 : ph0;
        const int ry = (int)(ph >> 8), rx = (int)(ph & 255u);
        const bool ok = pvalid && ph != 0xffffffffu;
        const unsigned off = ok ? (unsigned)(((qy * a.s + ry) * Wo + qx * a.s + rx) * (CH * 4)) : kOOB;
#pragma unroll
        for (int c4 = 0; c4 < CH; c4 += 4) buf_store(vs, f32x4{o[c4], o[c4 + 1], o[c4 + 2], o[c4 + 3]}, off, c4 * 4);
      }
    }
    if (!more) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// wpack[gstep][row][16] in the ring's LDS image order
__global__ void __launch_bounds__(256) syn_pack_kernel(const float* __restrict__ w, const SynUnit* __restrict__ units, int nunits,
                                                        SynGeom G, int nslab, int total_steps, float* __restrict__ wpack) {
  const size_t total = (size_t)total_steps * kUnit;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gstep = (int)(idx / kUnit);
    const int rem = (int)(idx - (size_t)gstep * kUnit);
    const int R = rem >> 4, pos = rem & 15;
    const int k16 = (((pos >> 2) ^ ((R >> 2) & 3)) << 2) + (pos & 3);
    int u = 0;
    for (int i = 1; i < nunits; ++i)
      if (gstep >= units[i].step0) u = i;           // units are stored in stream order (step0 ascending)
    const SynUnit U = units[u];
    const int local = gstep - U.step0;
    const int cc = local / U.ns, j = local - cc * U.ns;
    wpack[idx] = pack_value(G, U, cc, j, R, k16, w);
  }
}

__global__ void syn_tables_kernel(const float* b1, const float* beta, const float* gamma, int cp, int ch, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cp) out[i] = b1 ? b1[i] : 0.0f;
  else if (i < cp + ch) out[i] = beta ? beta[i - cp] : 1.0f;
  else if (i < cp + ch + ch * ch) out[i] = gamma ? gamma[i - cp - ch] : 0.0f;
}

// ---- host: the units of a layer
struct HostPlan {
  SynGeom G{};
  int ch = 0, has_res = 0;
  std::vector<SynUnit> units;     // stream order == processing order (most expensive first)
  int total_steps = 0;
  int nslab = 0;
};

static bool dim_ok(int k, int s, int pt) {
  for (int r = 0; r < s; ++r) {
    const int phi = (r + pt) % s, qo = (r + pt) / s, cnt = (k - phi + s - 1) / s;
    if (cnt < 1 || qo > 1 || qo - cnt + 1 < -1) return false;
  }
  return true;
}

static unsigned phase_mask(const SynGeom& G, int ry, int rx) {
  unsigned m = 0;
  for (int si = 0; si < 9; ++si) {
    const int dy = 1 - si / 3, dx = 1 - si % 3;
    const int phy = (ry + G.pt) % G.s, qoy = (ry + G.pt) / G.s, cy = (G.k - phy + G.s - 1) / G.s;
    const int phx = (rx + G.pt) % G.s, qox = (rx + G.pt) / G.s, cx = (G.k - phx + G.s - 1) / G.s;
    const int jy = qoy - dy, jx = qox - dx;
    if (jy >= 0 && jy < cy && jx >= 0 && jx < cx) m |= 1u << si;
  }
  return m;
}

static int popc(unsigned v) { return __builtin_popcount(v); }

static int build_host_plan(int k, int s, int cin, int ch, int has_res, HostPlan* P) {
  const int cp = ch * (has_res ? 2 : 1);
  if (k < s || !dim_ok(k, s, (k - s) / 2)) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn: kernel / stride outside the nine-shift frame");
  if (cp != 12 && cp != 24 && cp != 48) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn: 12, 24 or 48 output columns per phase");
  if (has_res && cp == 12) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn: 6 + 6 residual channels have no kernel instance");
  if (cin < 16 || cin % 16) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn: input channels must be a multiple of 16");
  if (s > 255) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn: stride");
  P->G = SynGeom{k, s, (k - s) / 2, cin, cp};
  P->ch = ch;
  P->has_res = has_res;
  P->nslab = cin / 16;
  const int nslot = 2 * (48 / cp);
  struct Ph { int ry, rx; unsigned mask; };
  std::vector<Ph> all;
  for (int ry = 0; ry < s; ++ry)
    for (int rx = 0; rx < s; ++rx) all.push_back({ry, rx, phase_mask(P->G, ry, rx)});
  // classes by shift set, larger sets first
  std::vector<unsigned> classes;
  for (auto& p : all)
    if (std::find(classes.begin(), classes.end(), p.mask) == classes.end()) classes.push_back(p.mask);
  std::stable_sort(classes.begin(), classes.end(), [](unsigned x, unsigned y) { return popc(x) > popc(y); });
  std::vector<std::vector<Ph>> units;
  std::vector<Ph> left;
  for (unsigned cm : classes) {
    std::vector<Ph> mem;
    for (auto& p : all)
      if (p.mask == cm) mem.push_back(p);
    size_t i = 0;
    for (; i + nslot <= mem.size(); i += nslot) units.emplace_back(mem.begin() + i, mem.begin() + i + nslot);
    for (; i < mem.size(); ++i) left.push_back(mem[i]);
  }
  // leftovers: start a unit with the largest shift set, fill with what grows the union least (ties: the larger set)
  while (!left.empty()) {
    std::vector<Ph> un{left.front()};
    left.erase(left.begin());
    unsigned uni = un[0].mask;
    while ((int)un.size() < nslot && !left.empty()) {
      size_t best = 0;
      int bg = 100, bp = -1;
      for (size_t i = 0; i < left.size(); ++i) {
        const int grow = popc(uni | left[i].mask) - popc(uni), pc = popc(left[i].mask);
        if (grow < bg || (grow == bg && pc > bp)) { best = i; bg = grow; bp = pc; }
      }
      uni |= left[best].mask;
      un.push_back(left[best]);
      left.erase(left.begin() + best);
    }
    units.push_back(un);
  }
  std::vector<SynUnit> out;
  for (auto& un : units) {
    SynUnit U{};
    unsigned uni = 0;
    for (auto& p : un) uni |= p.mask;
    U.ns = popc(uni);
    int j = 0;
    for (int si = 0; si < 9; ++si)
      if (uni >> si & 1u) {
        if (j < 8) U.sl0 |= (unsigned)si << (4 * j);
        else U.sl1 |= (unsigned)si << (4 * (j - 8));
        ++j;
      }
    // slots: the phases with the largest shift sets first, into the slots of the FIRST tiles (slot-in-half ascending, the two
    // lane halves alternating), so that the steps only the large sets use find the unit's third tile empty
    std::stable_sort(un.begin(), un.end(), [](const Ph& x, const Ph& y) { return popc(x.mask) > popc(y.mask); });
    const int sph = 48 / cp;
    unsigned slot_mask[kMaxSlots];
    for (int q = 0; q < kMaxSlots; ++q) { U.ph[q] = 0xffffffffu; slot_mask[q] = 0; }
    for (int i = 0; i < (int)un.size(); ++i) {
      const int q = (i & 1) * sph + (i >> 1);
      U.ph[q] = ((unsigned)un[i].ry << 8) | (unsigned)un[i].rx;
      slot_mask[q] = un[i].mask;
    }
    unsigned tile_mask[kNT] = {};                 // shifts some row of the tile uses
    for (int R = 0; R < kRows; ++R) {
      const int jt = R >> 5, i = R & 31, hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), vi = 16 * jt + r;
      tile_mask[jt] |= slot_mask[hh * sph + vi / cp];
    }
    U.cost = 0;
    for (int jj = 0; jj < U.ns; ++jj) {
      const int si = unit_shift(U, jj);
      const bool partial = !(tile_mask[kNT - 1] >> si & 1u);
      if (partial) U.pm |= 1u << jj;
      U.cost += partial ? kNT - 1 : kNT;
    }
    out.push_back(U);
  }
  std::stable_sort(out.begin(), out.end(), [](const SynUnit& x, const SynUnit& y) { return x.cost > y.cost; });
  int step = 0;
  for (auto& U : out) {
    U.step0 = step;
    step += U.ns * P->nslab;
  }
  P->units = out;
  P->total_steps = step;
  return SNTC_OK;
}

constexpr int kMaxDev = 16;
struct SynDevice {
  std::once_flag once;
  int rc = SNTC_OK;
  int num_cus = 0;
};
SynDevice g_syndev[kMaxDev];

template <int CP, bool RES>
static int set_attr() {
  SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&syn_kernel<CP, RES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)SynCfg<CP, RES>::LDS));
  return SNTC_OK;
}

static int syn_init(int* num_cus) {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the synthesis tables");
  SynDevice& D = g_syndev[dev];
  std::call_once(D.once, [&] {
    D.rc = [&]() -> int {
      hipDeviceProp_t prop;
      SNTC_HIP(hipGetDeviceProperties(&prop, dev));
      D.num_cus = prop.multiProcessorCount;
      if (int rc = set_attr<12, false>()) return rc;
      if (int rc = set_attr<24, false>()) return rc;
      if (int rc = set_attr<24, true>()) return rc;
      if (int rc = set_attr<48, false>()) return rc;
      if (int rc = set_attr<48, true>()) return rc;
      return SNTC_OK;
    }();
  });
  *num_cus = D.num_cus;
  return D.rc;
}

}  // namespace syn
}  // namespace sntc

using namespace sntc;
using namespace sntc::syn;

struct sntc_syn_plan {
  HostPlan P;
  int act_kind = 0;
  float* wpack = nullptr;
  SynUnit* units = nullptr;
  float* tables = nullptr;
  int max_workgroups = 0;
};

static void syn_free(sntc_syn_plan* p) {
  if (p->wpack) (void)hipFree(p->wpack);
  if (p->units) (void)hipFree(p->units);
  if (p->tables) (void)hipFree(p->tables);
  delete p;
}

static int syn_pack(sntc_syn_plan* p, const float* w1, const float* b1, const float* beta, const float* gamma, hipStream_t s) {
  const HostPlan& P = p->P;
  const size_t total = (size_t)P.total_steps * kUnit;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(syn_pack_kernel, dim3(blocks), dim3(256), 0, s, w1, p->units, (int)P.units.size(), P.G, P.nslab, P.total_steps, p->wpack);
  const int nt = P.G.cp + P.ch + P.ch * P.ch;
  hipLaunchKernelGGL(syn_tables_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, b1, beta, gamma, P.G.cp, P.ch, p->tables);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "synthesis weight packing");
  return SNTC_OK;
}

extern "C" int sntc_syn_supported(int k, int stride, int cin, int ch, int has_res) {
  HostPlan P;
  if (k < 1 || stride < 1 || ch < 1) return 0;
  return build_host_plan(k, stride, cin, ch, has_res ? 1 : 0, &P) == SNTC_OK ? 1 : 0;
}

extern "C" int sntc_syn_plan_create(int k, int stride, int cin, int ch, int has_res, int act_kind, const float* w1, const float* b1,
                                    const float* beta, const float* gamma, void* stream, sntc_syn_plan** plan) {
  if (!plan || !w1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_plan_create: null argument");
  if (act_kind < 0 || act_kind > 4) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn_plan_create: unknown activation");
  if ((act_kind == 1 || act_kind == 2) && (!beta || !gamma)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_plan_create: GDN parameters missing");
  if (k < 1 || stride < 1 || ch < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_plan_create: bad sizes");
  auto* p = new sntc_syn_plan();
  if (int rc = build_host_plan(k, stride, cin, ch, has_res ? 1 : 0, &p->P)) { delete p; return rc; }
  int cus = 0;
  if (int rc = syn_init(&cus)) { delete p; return rc; }
  p->act_kind = act_kind;
  const HostPlan& P = p->P;
  const int nt = P.G.cp + P.ch + P.ch * P.ch;
  if (hipMalloc(&p->wpack, sizeof(float) * (size_t)P.total_steps * kUnit) != hipSuccess ||
      hipMalloc(&p->units, sizeof(SynUnit) * P.units.size()) != hipSuccess || hipMalloc(&p->tables, sizeof(float) * nt) != hipSuccess) {
    syn_free(p);
    return fail(SNTC_ERR_HIP, "sntc_syn_plan_create: out of device memory");
  }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(p->units, P.units.data(), sizeof(SynUnit) * P.units.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    syn_free(p);
    return fail(SNTC_ERR_HIP, "sntc_syn_plan_create: unit table upload");
  }
  if (int rc = syn_pack(p, w1, b1, beta, gamma, s)) { syn_free(p); return rc; }
  *plan = p;
  return SNTC_OK;
}

extern "C" int sntc_syn_plan_update(sntc_syn_plan* p, const float* w1, const float* b1, const float* beta, const float* gamma, void* stream) {
  if (!p || !w1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_plan_update: null argument");
  if ((p->act_kind == 1 || p->act_kind == 2) && (!beta || !gamma)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_plan_update: GDN parameters missing");
  return syn_pack(p, w1, b1, beta, gamma, (hipStream_t)stream);
}

extern "C" void sntc_syn_plan_destroy(sntc_syn_plan* p) {
  if (p) syn_free(p);
}

extern "C" int sntc_syn_plan_set_workgroups(sntc_syn_plan* p, int max_workgroups) {
  if (!p || max_workgroups < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_plan_set_workgroups: bad argument");
  p->max_workgroups = max_workgroups;
  return SNTC_OK;
}

extern "C" int64_t sntc_syn_flops(const sntc_syn_plan* p, int64_t latent_pixels) {
  if (!p || latent_pixels < 0) return -1;
  return 2 * latent_pixels * p->P.G.k * p->P.G.k * p->P.G.cin * p->P.G.cp;
}

extern "C" int64_t sntc_syn_workspace_bytes(const sntc_syn_plan* p) { return p ? 256 : -1; }

extern "C" int sntc_syn_forward(const sntc_syn_plan* p, const sntc_syn_batch* batches, int nbatches, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (!p || !batches || !workspace) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: null argument");
  if (nbatches < 1 || nbatches > kMaxSynGroups) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: 1 .. 4 batches per call");
  if (workspace_bytes < 256) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: workspace smaller than sntc_syn_workspace_bytes()");
  const HostPlan& P = p->P;
  SynArgs a{};
  int64_t tiles = 0;
  int ng = 0;
  for (int i = 0; i < nbatches; ++i) {
    const sntc_syn_batch& b = batches[i];
    if (b.n < 0 || b.h < 0 || b.w < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: negative size");
    if (b.n == 0 || b.h == 0 || b.w == 0) continue;
    if (!b.y_hat || !b.hidden) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: null tensor");
    if (kTileM + 2 * b.w + 2 > kPPMax) return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn_forward: latent rows wider than 127 pixels");
    const int64_t hw = (int64_t)b.h * b.w;
    if (hw * P.G.cin * 4 >= (1LL << 31) || hw * P.G.s * P.G.s * P.ch * 4 >= (1LL << 31))
      return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: one image's tensor of 2 GiB or more");
    SynGroup& g = a.g[ng++];
    g.x = b.y_hat; g.v = b.hidden; g.n = b.n; g.h = b.h; g.w = b.w;
    g.tile0 = (int)tiles;
    g.tpi = (int)((hw + kTileM - 1) / kTileM);
    tiles += (int64_t)b.n * g.tpi;
    if (tiles * (int64_t)P.units.size() >= (1LL << 30)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_forward: too many work items");
  }
  if (ng == 0) return SNTC_OK;
  int cus = 0;
  if (int rc = syn_init(&cus)) return rc;
  a.wpack = p->wpack; a.units = p->units; a.tables = p->tables;
  a.queue = reinterpret_cast<int*>(workspace);
  a.wbytes = (unsigned)((size_t)P.total_steps * kUnit * 4);
  a.nunits = (int)P.units.size(); a.ntiles = (int)tiles; a.nitems = a.nunits * a.ntiles;
  a.cin = P.G.cin; a.nslab = P.nslab; a.s = P.G.s; a.act_kind = p->act_kind;
  a.ngroups = ng;
#ifdef SNTC_DIAG
  if (const char* e = getenv("SNTC_SYN_DBG")) a.dbg = atoi(e);     // diagnostic builds only (make DIAG=1): results are WRONG with it
#endif
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(workspace, 64, s)) return zrc;
  const int grid = std::min<int64_t>(a.nitems, p->max_workgroups > 0 ? p->max_workgroups : cus);
  const void* fn = nullptr;
  size_t lds = 0;
  const int cp = P.G.cp;
  if (cp == 12 && !P.has_res) { fn = reinterpret_cast<const void*>(&syn_kernel<12, false>); lds = SynCfg<12, false>::LDS; }
  else if (cp == 24 && !P.has_res) { fn = reinterpret_cast<const void*>(&syn_kernel<24, false>); lds = SynCfg<24, false>::LDS; }
  else if (cp == 24 && P.has_res) { fn = reinterpret_cast<const void*>(&syn_kernel<24, true>); lds = SynCfg<24, true>::LDS; }
  else if (cp == 48 && !P.has_res) { fn = reinterpret_cast<const void*>(&syn_kernel<48, false>); lds = SynCfg<48, false>::LDS; }
  else if (cp == 48 && P.has_res) { fn = reinterpret_cast<const void*>(&syn_kernel<48, true>); lds = SynCfg<48, true>::LDS; }
  else return fail(SNTC_ERR_UNSUPPORTED, "sntc_syn_forward: no kernel for this width");
  void* params[] = {&a};
  hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(512), params, lds, s);
  if (e != hipSuccess) return hip_fail(e, "synthesis launch");
  return SNTC_OK;
}

// The decomposition checked on the host, no device involved (CPU test tier): the unit tables and pack_value() drive a plain
// loop nest over (tile pixel, unit, slab, step, row) exactly as the kernel walks them -- shifted sources, zero outside the
// image, slot -> phase -> output pixel -- and the result is compared with the scatter form of Conv2DTranspose(SAME)
// (SURVEY.md App. A.2).  Returns the largest absolute difference through *max_err (doubles: only the index arithmetic is on trial).
extern "C" int sntc_syn_selfcheck(int k, int stride, int cin, int ch, int has_res, int h, int w, unsigned seed, double* max_err) {
  if (!max_err) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_selfcheck: null argument");
  HostPlan P;
  if (int rc = build_host_plan(k, stride, cin, ch, has_res ? 1 : 0, &P)) return rc;
  if (h < 1 || w < 1 || kTileM + 2 * w + 2 > kPPMax) return fail(SNTC_ERR_BAD_SHAPE, "sntc_syn_selfcheck: bad image size");
  const SynGeom G = P.G;
  const int cp = G.cp, s = G.s;
  unsigned st = seed * 2654435761u + 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (double)((st >> 8) & 0xffff) / 32768.0 - 1.0; };
  std::vector<float> wk((size_t)k * k * cp * cin), x((size_t)h * w * cin);
  for (auto& v : wk) v = (float)rnd();
  for (auto& v : x) v = (float)rnd();
  const int Ho = h * s, Wo = w * s;
  std::vector<double> ref((size_t)Ho * Wo * cp, 0.0), got((size_t)Ho * Wo * cp, 0.0);
  std::vector<int> hits((size_t)Ho * Wo, 0);
  for (int iy = 0; iy < h; ++iy)
    for (int ix = 0; ix < w; ++ix)
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          const int oy = iy * s + ky - G.pt, ox = ix * s + kx - G.pt;
          if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
          for (int c = 0; c < cp; ++c) {
            double acc = 0.0;
            for (int ci = 0; ci < cin; ++ci) acc += (double)x[((size_t)iy * w + ix) * cin + ci] * wk[((size_t)(ky * k + kx) * cp + c) * cin + ci];
            ref[((size_t)oy * Wo + ox) * cp + c] += acc;
          }
        }
  const int sph = 48 / cp;
  const int HW = h * w;
  for (const SynUnit& U : P.units)
    for (int m = 0; m < HW; ++m) {
      const int qy = m / w, qx = m % w;
      double acc[kRows];
      for (int R = 0; R < kRows; ++R) acc[R] = 0.0;
      for (int cc = 0; cc < P.nslab; ++cc)
        for (int j = 0; j < U.ns; ++j) {
          const int si = unit_shift(U, j), dy = 1 - si / 3, dx = 1 - si % 3;
          const int sy = qy + dy, sx = qx + dx;
          if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;
          for (int R = 0; R < kRows; ++R)
            for (int k16 = 0; k16 < 16; ++k16) {
              const float wv = pack_value(G, U, cc, j, R, k16, wk.data());
              if ((U.pm >> j & 1u) && R >= 32 * (kNT - 1)) {        // a partial step: the kernel leaves the third tile out
                if (wv != 0.0f) { *max_err = 1e30; return SNTC_OK; }
                continue;
              }
              acc[R] += (double)wv * x[((size_t)sy * w + sx) * cin + cc * 16 + k16];
            }
        }
      for (int R = 0; R < kRows; ++R) {
        const int jt = R >> 5, i = R & 31, hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), vi = 16 * jt + r;
        const unsigned ph = U.ph[hh * sph + vi / cp];
        if (ph == 0xffffffffu) {
          if (acc[R] != 0.0) { *max_err = 1e30; return SNTC_OK; }
          continue;
        }
        const int oy = qy * s + (int)(ph >> 8), ox = qx * s + (int)(ph & 255u);
        got[((size_t)oy * Wo + ox) * cp + vi % cp] += acc[R];
        if (vi % cp == 0) hits[(size_t)oy * Wo + ox] += 1;
      }
    }
  double e = 0.0;
  for (size_t i = 0; i < ref.size(); ++i) e = std::max(e, std::fabs(ref[i] - got[i]));
  for (int v : hits)
    if (v != 1) e = 1e30;          // every output pixel is produced by exactly one (unit, slot)
  *max_err = e;
  return SNTC_OK;
}

// the units a layer would get, without a device (tests, tools): same records as sntc_syn_plan_units
extern "C" int sntc_syn_describe(int k, int stride, int cin, int ch, int has_res, int* out, int capacity) {
  HostPlan P;
  if (k < 1 || stride < 1 || ch < 1 || (!out && capacity > 0)) return -1;
  if (build_host_plan(k, stride, cin, ch, has_res ? 1 : 0, &P) != SNTC_OK) return -1;
  const int n = (int)P.units.size();
  for (int i = 0; i < n && i < capacity; ++i) {
    const SynUnit& U = P.units[i];
    int nph = 0;
    for (int q = 0; q < kMaxSlots; ++q) nph += U.ph[q] != 0xffffffffu;
    out[4 * i] = U.ns; out[4 * i + 1] = nph | (U.cost << 8) | ((int)U.pm << 16); out[4 * i + 2] = (int)U.sl0; out[4 * i + 3] = (int)U.sl1;
  }
  return n;
}

// the plan's units for tests and tools: writes up to `capacity` records of 4 ints (steps per slab, phases, packed shifts lo, hi)
extern "C" int sntc_syn_plan_units(const sntc_syn_plan* p, int* out, int capacity) {
  if (!p || (!out && capacity > 0)) return -1;
  const int n = (int)p->P.units.size();
  for (int i = 0; i < n && i < capacity; ++i) {
    const SynUnit& U = p->P.units[i];
    int nph = 0;
    for (int q = 0; q < kMaxSlots; ++q) nph += U.ph[q] != 0xffffffffu;
    out[4 * i] = U.ns; out[4 * i + 1] = nph | (U.cost << 8) | ((int)U.pm << 16); out[4 * i + 2] = (int)U.sl0; out[4 * i + 3] = (int)U.sl1;
  }
  return n;
}

// conv_plan.hip -- host side of the convolution entry points of sntc.h: turns a Keras / TFC layer
// description into gather-GEMM groups, packs the weights on the device once, and launches.
//
// Phase-grouped transposed convolution: for Conv2DTranspose(k, s, SAME) with
// pt = max(k-s,0)//2, output row oy gathers  oy + pt = qy*s + phi,  ky = phi + jy*s,  i = qy - jy.
// ceil((k-phi)/s) takes at most two values over phi in [0,s), so the s*s output phases fall into
// <= 4 groups that share a tap pattern; each group is one GEMM whose columns are
// (phase, channel) pairs: N = phases*Cout, K = taps*Cin -- no zero stuffing, no col2im.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <cstdlib>
#include <vector>
#include "sntc_internal.h"

namespace sntc {

// One packed element: column `col`, position `k` of a group's [Ncol][K] weight rows (K order and kernel layouts below).
struct PackDesc {
  const float* w;
  float* wp;
  const int* taps;
  const unsigned* cols;
  int T, Cin, Cout, K, Ncol, kw, s, pt, pl, phase_mode, slab_major, out_major, bf3, rowpack;
};

__device__ __forceinline__ void pack_element(const PackDesc& d, size_t idx) {
  const int col = (int)(idx / d.K);
  const int k = (int)(idx - (size_t)col * d.K);
  float v = 0.0f;
  if (d.rowpack) {
    // row-packed first layer: K stage t = kernel row ky, its 16 slots = the kw * Cin values (kx, c) of that row, then zeros
    const int t = k >> 4, ci = k & 15;
    if (t < d.T && ci < d.kw * d.Cin) {
      const int kx = ci / d.Cin, c = ci - kx * d.Cin;
      v = d.w[((size_t)(t * d.kw + kx) * d.Cin + c) * d.Cout + (d.cols[col] & 0xffff)];
    }
  } else if (k < d.T * d.Cin) {
    // K order (csrc/gather_gemm.hip): Cin % 16 == 0 -> channel slab outermost, k = cc * T * 16 + t * 16 + c;
    // otherwise (dword gather path) tap-major, k = t * Cin + ci
    int t, ci;
    if (d.slab_major) {
      const int cc = k / (d.T * 16);
      const int r = k - cc * d.T * 16;
      t = r >> 4;
      ci = cc * 16 + (r & 15);
    } else {
      t = k / d.Cin;
      ci = k - t * d.Cin;
    }
    const int tap = d.taps[t];
    const unsigned ce = d.cols[col];
    const int ch = ce & 0xffff;
    int ky = tap >> 16, kx = tap & 0xffff;
    if (d.phase_mode) {
      ky = ((int)((ce >> 24) & 0xff) - 128 + d.pt) + ky * d.s;
      kx = ((int)((ce >> 16) & 0xff) - 128 + d.pl) + kx * d.s;
    }
    const size_t tapi = (size_t)ky * d.kw + kx;
    v = d.out_major ? d.w[(tapi * d.Cout + ch) * d.Cin + ci] : d.w[(tapi * d.Cin + ci) * d.Cout + ch];
  }
  if (d.bf3) {           // three bf16 planes, 96 B per (column, 16-deep stage): [plane][16]
    __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    __bf16 mid = (__bf16)r1;
    __bf16 lo = (__bf16)(r1 - (float)mid);
    __bf16* o = reinterpret_cast<__bf16*>(d.wp) + ((size_t)col * (d.K / 16) + (k >> 4)) * 48 + (k & 15);
    o[0] = hi; o[16] = mid; o[32] = lo;
  } else {
    d.wp[idx] = v;
  }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                    const int* __restrict__ taps, const unsigned* __restrict__ cols,
                                    int T, int Cin, int Cout, int K, int Ncol, int kind, int kw, int s,
                                    int pt, int pl, int phase_mode, int slab_major, int out_major, int bf3, int rowpack) {
  const PackDesc d{w, wp, taps, cols, T, Cin, Cout, K, Ncol, kw, s, pt, pl, phase_mode, slab_major, out_major, bf3, rowpack};
  const size_t total = (size_t)Ncol * K;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x)
    pack_element(d, idx);
}

__device__ __forceinline__ void pack_fused_w2_element(const float* wp, float* w2f, int K, int Ncol, int i) {
  const int e = i & 3, lane = (i >> 2) & 63, Q = (i >> 8) % 12, t = ((i >> 8) / 12) % 3, hf = (i >> 8) / 36;
  const int k = 8 * Q + 4 * (lane >> 5) + e, n = 96 * hf + 32 * t + (lane & 31);
  w2f[i] = (k < K && n < Ncol) ? wp[(size_t)n * K + k] : 0.0f;
}

// Every plan of a training step re-packed by ONE launch (sntc_plan_group_update): job j is a weight group of a plan
// (kind 0), a bias copy (kind 1: wp[i] = w[i], Ncol floats) or a fused-tail fragment copy (kind 2, runs in a second
// launch because it reads what kind 0 wrote); block b works on elements [first[b], first[b] + kPackChunk) of job[b].
constexpr int kPackChunk = 4096;
struct PackJob {
  PackDesc d;
  int kind;
};

__global__ void __launch_bounds__(256) pack_group_kernel(const PackJob* __restrict__ jobs, const int* __restrict__ blk_job,
                                                         const long long* __restrict__ blk_first) {
  const PackJob& J = jobs[blk_job[blockIdx.x]];
  const size_t total = J.kind == 0 ? (size_t)J.d.Ncol * J.d.K : J.kind == 1 ? (size_t)J.d.Ncol : (size_t)(2 * 3 * 12 * 64 * 4);
  const size_t lo = (size_t)blk_first[blockIdx.x];
  const size_t hi = lo + kPackChunk < total ? lo + kPackChunk : total;
  for (size_t idx = lo + threadIdx.x; idx < hi; idx += 256) {
    if (J.kind == 0) pack_element(J.d, idx);
    else if (J.kind == 1) J.d.wp[idx] = J.d.w[idx];
    else pack_fused_w2_element(J.d.w, J.d.wp, J.d.K, J.d.Ncol, (int)idx);
  }
}

// W2 of a fused ResidualBlock tail, from the 1x1 plan's packed [Ncol = 192][K = 96] rows into the order the FUSE2 instance
// copies to LDS and reads as MFMA B fragments: [half hf][column tile t][k-quad pair Q][lane][e] = W2[k = 8 Q + 4 (lane / 32) + e]
// [n = 96 hf + 32 t + lane % 32]
__global__ void pack_fused_w2_kernel(const float* __restrict__ wp, float* __restrict__ w2f, int K, int Ncol) {
  const int total = 2 * 3 * 12 * 64 * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
    pack_fused_w2_element(wp, w2f, K, Ncol, i);
}

}  // namespace sntc

using namespace sntc;

struct sntc_conv_plan {
  sntc_conv_desc d;
  int device = 0;
  bool up = false;          // transposed family
  bool phase_mode = false;  // up && stride > 1
  int pt = 0, pl = 0;       // fixed pads (transposed / SignalConv2D); Keras Conv2D SAME is per call
  int ngroups = 0;
  bool vec = false;
  bool exact_grid = false;  // phase mode: every group has one grid origin -> macro grid == input grid (no +1 row/col)
  struct Grp {
    int T = 0, K = 0, Ncol = 0;
    int tw = 1;               // taps form a dense th x tw grid in row-major order (checked when the plan is built)
    int q0y = 0, q0x = 0;
    float* wp = nullptr;
    int* taps = nullptr;
    unsigned* cols = nullptr;
  } g[kMaxGroups];
  float* bias = nullptr;
  int tile = 0;             // forced gather-GEMM tile variant of THIS plan (0 = heuristic): profiling / tests only
  bool bf3 = false;         // desc.reserved[1]: bf16 x 3 split precision (weights packed as three bf16 planes)
  bool rowpack = false;     // desc.reserved[2] == 1: small-Cin forward convolution on a caller-padded input, one kernel ROW per K stage
  bool s3 = false;          // desc.reserved[1] == 2: the INPUT arrives pre-split too (format S3, 6 B per element): csrc/bf3_gemm.hip
  bool out_major = false;   // kernel array is [kh, kw, Cout, Cin] (Keras Conv2DTranspose; any kind with desc.kernel_io_swapped)
  int dma = -1;             // direct-to-LDS staging: -1 default (kDefaultDma), 0 off, 1 on (sntc_conv_plan_set_schedule bit 1)
  bool no_stream_k = false; // force the static one-workgroup-per-tile schedule (tests: both schedules give identical bits)
  bool force_stream_k = false;  // ignore the short-tile rule: stream-K wherever the launch is large enough (tests)
  int colm = -1;            // fp32 stream-K unit order: -1 rule (column tile outermost where one group's packed weights exceed an
                            // XCD's 4 MB L2), 0 strip-major, 1 column-major wherever the twin exists (sntc_conv_plan_set_schedule bits 5, 6)
  bool no_halo = false;     // pre-split plans: stage every tap's activation rows separately even where one patch per slab would do (A/B)
  float* w2f = nullptr;     // this 1x1 plan's weights in the fused ResidualBlock tail's fragment order (fusable_second plans only)
  // sntc_conv_plan_tune: (n, h, w) -> measured best (tile variant, schedule).  Every candidate computes the same k-ordered chains,
  // so a tuned entry changes the launch's speed and never its bits.  Guarded: plans are shared by concurrent streams / threads.
  struct Choice { int variant = 0; int sk = 0; };
  mutable std::mutex tune_mu;
  std::map<std::array<int, 3>, Choice> tuned;
};
typedef sntc_conv_plan::Choice TuneChoice;

extern "C" int sntc_conv_plan_set_tile(sntc_conv_plan* p, int variant) {
  if (!p) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_set_tile: null plan");
  if (variant < 0 || variant > (p->s3 ? 13 : kNumVariants) || (p->s3 && variant != 0 && variant < 11))
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_plan_set_tile: unknown tile variant");
  p->tile = variant;
  return SNTC_OK;
}

// Direct-to-LDS staging is OFF by default.  Alone on the device it is worth +1 ... 3 % on long contractions (3x3 and larger
// kernels, tools/ab_dma.sh) and costs the short-K 1x1 layers a resident workgroup (fourth ring slot: 57 -> 48 TFLOP/s on
// 96 -> 192 + skip); but with two batch shapes of the Kodak set in flight on two streams (bench.py's default) its larger LDS
// footprint leaves no room for the other stream's workgroups and the decode step went from 3.43 to 3.75 ms.  Bit-identical
// either way (tests/test_hip_fullsize.py); kept selectable per plan.
static bool plan_dma(const sntc_conv_plan* p) {
  return p->dma > 0 && p->vec && p->d.prologue == SNTC_PRO_NONE && !p->bf3;
}

extern "C" int sntc_conv_plan_set_schedule(sntc_conv_plan* p, int flags) {
  if (!p) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_set_schedule: null plan");
  p->no_stream_k = (flags & 1) == 0;
  p->force_stream_k = (flags & 8) != 0;
  p->no_halo = (flags & 16) != 0;
  p->colm = (flags & 32) ? ((flags & 64) ? 1 : 0) : -1;   // bit 5: "bit 6 is meaningful"; bit 6: column-major stream-K unit order on / off
  p->dma = (flags & 4) ? ((flags & 2) ? 1 : 0) : -1;      // bit 2: "bit 1 is meaningful"; bit 1: direct-to-LDS staging on / off
  return SNTC_OK;
}

extern "C" void sntc_conv_plan_destroy(sntc_conv_plan* p) {
  if (!p) return;
  for (int i = 0; i < kMaxGroups; ++i) {
    if (p->g[i].wp) (void)hipFree(p->g[i].wp);
    if (p->g[i].taps) (void)hipFree(p->g[i].taps);
    if (p->g[i].cols) (void)hipFree(p->g[i].cols);
  }
  if (p->bias) (void)hipFree(p->bias);
  if (p->w2f) (void)hipFree(p->w2f);
  delete p;
}

// A plan that can be the SECOND half of a fused ResidualBlock tail (1x1, 96 -> 192, no activation, fp32) keeps its weights
// in the fused kernel's fragment order as well.  Packed here -- at creation and at every update, on the caller's stream,
// like the plan's own weights -- so that the forward calls never write to a plan (plans are shared by concurrent streams).
static bool fusable_second(const sntc_conv_plan* p) {
  const sntc_conv_desc& b = p->d;
  return !p->up && !p->bf3 && b.kh == 1 && b.kw == 1 && b.stride == 1 && b.cin == 96 && b.cout == 192 &&
         b.prologue == SNTC_PRO_NONE && b.act == SNTC_ACT_NONE && p->ngroups == 1 && p->g[0].K == 96;
}

static int pack_fused_second(sntc_conv_plan* p, hipStream_t stream) {
  if (!fusable_second(p)) return SNTC_OK;
  if (!p->w2f) SNTC_HIP(hipMalloc(&p->w2f, sizeof(float) * 2 * 3 * 12 * 64 * 4));
  hipLaunchKernelGGL(pack_fused_w2_kernel, dim3(72), dim3(256), 0, stream, p->g[0].wp, p->w2f, p->g[0].K, p->g[0].Ncol);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

static int build_plan(sntc_conv_plan* p, const float* weight, const float* bias, hipStream_t stream) {
  const sntc_conv_desc& d = p->d;
  const int k = d.kh, s = d.stride;
  struct HostGrp { std::vector<int> taps; std::vector<unsigned> cols; int q0y = 0, q0x = 0; bool uniform = true; };
  std::vector<HostGrp> groups;
  auto enc = [](int oyo, int oxo, int ch) { return ((unsigned)(oyo + 128) << 24) | ((unsigned)(oxo + 128) << 16) | (unsigned)ch; };

  if (!p->phase_mode) {
    HostGrp hg;
    for (int ky = 0; ky < d.kh; ++ky)
      for (int kx = 0; kx < (p->rowpack ? 1 : d.kw); ++kx) hg.taps.push_back((ky << 16) | kx);
    for (int ch = 0; ch < d.cout; ++ch) hg.cols.push_back(enc(0, 0, ch));
    groups.push_back(std::move(hg));
  } else {
    // classes of phases per axis by tap count
    auto classes = [&](int kk, std::vector<std::pair<int, std::vector<int>>>& out) {
      for (int phi = 0; phi < s; ++phi) {
        const int cnt = (kk - phi + s - 1) / s;
        if (cnt <= 0) return false;
        bool found = false;
        for (auto& c : out)
          if (c.first == cnt) { c.second.push_back(phi); found = true; }
        if (!found) out.push_back({cnt, {phi}});
      }
      std::sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
      return true;
    };
    std::vector<std::pair<int, std::vector<int>>> cy, cx;
    if (!classes(d.kh, cy) || !classes(d.kw, cx))
      return fail(SNTC_ERR_UNSUPPORTED, "transposed convolution with kernel < stride is not supported");
    for (auto& a : cy)
      for (auto& b : cx) {
        HostGrp hg;
        for (int jy = 0; jy < a.first; ++jy)
          for (int jx = 0; jx < b.first; ++jx) hg.taps.push_back((jy << 16) | jx);
        for (int py : a.second)
          for (int px : b.second)
            for (int ch = 0; ch < d.cout; ++ch) hg.cols.push_back(enc(py - p->pt, px - p->pl, ch));
        // first valid macro row of a phase: oy = q s + phi - pt >= 0  ->  q0 = 1 iff phi < pt (pt < s here)
        auto q0 = [&](int phi, int pad) { return phi < pad ? 1 : 0; };
        hg.q0y = q0(a.second[0], p->pt);
        hg.q0x = q0(b.second[0], p->pl);
        for (int py : a.second) hg.uniform &= q0(py, p->pt) == hg.q0y;
        for (int px : b.second) hg.uniform &= q0(px, p->pl) == hg.q0x;
        groups.push_back(std::move(hg));
      }
    std::stable_sort(groups.begin(), groups.end(),
                     [](const HostGrp& a, const HostGrp& b) { return a.taps.size() > b.taps.size(); });
  }
  if ((int)groups.size() > kMaxGroups) return fail(SNTC_ERR_UNSUPPORTED, "too many phase groups");
  p->ngroups = (int)groups.size();
  p->exact_grid = p->phase_mode && p->pt < s && p->pl < s;
  for (auto& hg : groups) p->exact_grid = p->exact_grid && hg.uniform;
  for (int gi = 0; gi < p->ngroups; ++gi) {
    auto& hg = groups[gi];
    auto& G = p->g[gi];
    if (p->exact_grid) { G.q0y = hg.q0y; G.q0x = hg.q0x; }
    G.T = (int)hg.taps.size();
    G.Ncol = (int)hg.cols.size();
    G.tw = 1;
    for (int t = 0; t < G.T; ++t) G.tw = std::max(G.tw, (hg.taps[t] & 0xffff) + 1);
    for (int t = 0; t < G.T; ++t)     // the kernel walks taps arithmetically: (t / tw, t % tw)
      if (hg.taps[t] != (((t / G.tw) << 16) | (t % G.tw))) return fail(SNTC_ERR_UNSUPPORTED, "tap table is not a dense row-major grid");
    G.K = p->rowpack ? G.T * kStage : ((G.T * d.cin + kStage - 1) / kStage) * kStage;
    if ((int64_t)G.Ncol * G.K * 4 >= (1LL << 31)) return fail(SNTC_ERR_UNSUPPORTED, "packed weights of one phase group must be < 2 GiB");
    SNTC_HIP(hipMalloc(&G.taps, sizeof(int) * std::max(1, G.T)));
    SNTC_HIP(hipMalloc(&G.cols, sizeof(unsigned) * G.Ncol));
    SNTC_HIP(hipMalloc(&G.wp, (p->bf3 ? 6 : sizeof(float)) * (size_t)G.Ncol * std::max(kStage, G.K)));
    SNTC_HIP(hipMemcpyAsync(G.taps, hg.taps.data(), sizeof(int) * G.T, hipMemcpyHostToDevice, stream));
    SNTC_HIP(hipMemcpyAsync(G.cols, hg.cols.data(), sizeof(unsigned) * G.Ncol, hipMemcpyHostToDevice, stream));
    // the host vectors must outlive the async copies
    SNTC_HIP(hipStreamSynchronize(stream));
    const size_t total = (size_t)G.Ncol * G.K;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    if (total > 0)
      hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, weight, G.wp, G.taps, G.cols,
                         G.T, d.cin, d.cout, G.K, G.Ncol, d.kind, d.kw, s, p->pt, p->pl, p->phase_mode ? 1 : 0, p->vec ? 1 : 0, p->out_major ? 1 : 0, p->bf3 ? 1 : 0, p->rowpack ? 1 : 0);
    SNTC_HIP(hipGetLastError());
  }
  if (bias) {
    SNTC_HIP(hipMalloc(&p->bias, sizeof(float) * d.cout));
    SNTC_HIP(hipMemcpyAsync(p->bias, bias, sizeof(float) * d.cout, hipMemcpyDeviceToDevice, stream));
  }
  (void)k;
  return pack_fused_second(p, stream);
}

extern "C" int sntc_conv_plan_create(const sntc_conv_desc* desc, const float* weight, const float* bias,
                                     void* stream, sntc_conv_plan** plan) {
  if (!desc || !weight || !plan) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_create: null argument");
  const sntc_conv_desc& d = *desc;
  if (d.kind < SNTC_CONV2D || d.kind > SNTC_SIGNAL_UP) return fail(SNTC_ERR_UNSUPPORTED, "unknown conv kind");
  if (d.kh < 1 || d.kw < 1 || d.kh > 64 || d.kw > 64 || d.stride < 1 || d.stride > 64 || d.cin < 1 || d.cout < 1 ||
      d.cout > 65535)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_create: bad kernel / stride / channel sizes");
  if (d.act < SNTC_ACT_NONE || d.act > SNTC_ACT_SIGMOID || d.prologue < 0 || d.prologue > SNTC_PRO_SQUARE ||
      d.epilogue < 0 || d.epilogue > SNTC_EPI_MASK_LEAKY)
    return fail(SNTC_ERR_UNSUPPORTED, "unknown activation / prologue / epilogue");
  int rc = gg_init();
  if (rc) return rc;
  auto* p = new sntc_conv_plan();
  p->d = d;
  SNTC_HIP(hipGetDevice(&p->device));
  p->up = (d.kind == SNTC_CONV2D_TRANSPOSE || d.kind == SNTC_SIGNAL_UP);
  p->phase_mode = p->up && d.stride > 1;
  if (d.kind == SNTC_CONV2D_TRANSPOSE) {
    p->pt = std::max(d.kh - d.stride, 0) / 2;
    p->pl = std::max(d.kw - d.stride, 0) / 2;
  } else if (d.kind == SNTC_SIGNAL_UP) {
    p->pt = (d.kh - 1) / 2;
    p->pl = (d.kw - 1) / 2;
  } else if (d.kind == SNTC_SIGNAL_DOWN) {
    p->pt = d.kh / 2;
    p->pl = d.kw / 2;
  }
  p->rowpack = d.reserved[2] == 1;
  if (p->rowpack && (d.kind != SNTC_CONV2D || d.kw * d.cin > kStage || d.prologue != SNTC_PRO_NONE || d.reserved[1] != 0)) {
    delete p;
    return fail(SNTC_ERR_UNSUPPORTED, "row-packed plans: Keras Conv2D with kw * Cin <= 16 (the RGB first layer), fp32, no prologue");
  }
  p->vec = p->rowpack || (d.cin % kStage) == 0;
  // Keras Conv2DTranspose stores [kh, kw, Cout, Cin]; everything else [kh, kw, Cin, Cout] -- unless the caller says the
  // array is the channel-transposed one (the adjoint of a SignalConv2D layer runs on the layer's own kernel array)
  p->out_major = (d.kind == SNTC_CONV2D_TRANSPOSE) != (d.reserved[0] != 0);
  p->bf3 = d.reserved[1] != 0;
  p->s3 = d.reserved[1] == 2;
  if (p->bf3 && (!p->vec || d.prologue != SNTC_PRO_NONE)) {
    delete p;
    return fail(SNTC_ERR_UNSUPPORTED, "bf16 x 3 plans need Cin % 16 == 0 and no prologue");
  }
  if (p->s3 && ((d.cout & 3) || (d.epilogue != SNTC_EPI_STORE && d.epilogue != SNTC_EPI_ADD && d.epilogue != SNTC_EPI_GATE &&
                                d.epilogue != SNTC_EPI_MASK_RELU && d.epilogue != SNTC_EPI_MASK_LEAKY))) {
    delete p;
    return fail(SNTC_ERR_UNSUPPORTED, "pre-split bf16 x 3 plans need Cout % 4 == 0 and a store / add / gate / mask epilogue");
  }
  rc = build_plan(p, weight, bias, (hipStream_t)stream);
  if (rc) {
    sntc_conv_plan_destroy(p);
    return rc;
  }
  *plan = p;
  return SNTC_OK;
}

// Re-pack the weights (and bias) of an existing plan from new device arrays: the training step refreshes its forward
// and adjoint plans after every optimizer update; geometry, tap / column tables and tile choices are unchanged.
extern "C" int sntc_conv_plan_update(sntc_conv_plan* p, const float* weight, const float* bias, void* stream) {
  if (!p || !weight) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_update: null argument");
  if ((bias != nullptr) != (p->bias != nullptr)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_update: bias presence differs from the plan");
  hipStream_t s = (hipStream_t)stream;
  const sntc_conv_desc& d = p->d;
  for (int gi = 0; gi < p->ngroups; ++gi) {
    auto& G = p->g[gi];
    const size_t total = (size_t)G.Ncol * G.K;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    if (total > 0)
      hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, weight, G.wp, G.taps, G.cols, G.T, d.cin, d.cout, G.K,
                         G.Ncol, d.kind, d.kw, d.stride, p->pt, p->pl, p->phase_mode ? 1 : 0, p->vec ? 1 : 0, p->out_major ? 1 : 0, p->bf3 ? 1 : 0, p->rowpack ? 1 : 0);
    SNTC_HIP(hipGetLastError());
  }
  if (bias) SNTC_HIP(hipMemcpyAsync(p->bias, bias, sizeof(float) * d.cout, hipMemcpyDeviceToDevice, s));
  return pack_fused_second(p, s);
}

// ---- plan group: every plan of a model re-packed by one launch (training step) ----
struct sntc_plan_group {
  PackJob* jobs = nullptr;          // device
  int* blk_job = nullptr;
  long long* blk_first = nullptr;
  int nblocks = 0;                  // blocks of the weight / bias jobs
  int nblocks2 = 0;                 // blocks of the fused-tail fragment jobs (second launch: they read the first one's output)
};

extern "C" void sntc_plan_group_destroy(sntc_plan_group* g) {
  if (!g) return;
  if (g->jobs) (void)hipFree(g->jobs);
  if (g->blk_job) (void)hipFree(g->blk_job);
  if (g->blk_first) (void)hipFree(g->blk_first);
  delete g;
}

extern "C" int sntc_plan_group_create(sntc_conv_plan* const* plans, const float* const* weights, const float* const* biases,
                                      int count, void* stream, sntc_plan_group** out) {
  if (!plans || !weights || !biases || !out || count < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_plan_group_create: null argument");
  std::vector<PackJob> jobs, jobs2;
  for (int i = 0; i < count; ++i) {
    sntc_conv_plan* p = plans[i];
    if (!p || !weights[i]) return fail(SNTC_ERR_BAD_SHAPE, "sntc_plan_group_create: null plan or weight array");
    if ((biases[i] != nullptr) != (p->bias != nullptr))
      return fail(SNTC_ERR_BAD_SHAPE, "sntc_plan_group_create: bias presence differs from the plan");
    const sntc_conv_desc& d = p->d;
    for (int gi = 0; gi < p->ngroups; ++gi) {
      auto& G = p->g[gi];
      PackJob j{};
      j.d = PackDesc{weights[i], G.wp, G.taps, G.cols, G.T, d.cin, d.cout, G.K, G.Ncol, d.kw, d.stride, p->pt, p->pl,
                     p->phase_mode ? 1 : 0, p->vec ? 1 : 0, p->out_major ? 1 : 0, p->bf3 ? 1 : 0, p->rowpack ? 1 : 0};
      j.kind = 0;
      if ((size_t)G.Ncol * G.K > 0) jobs.push_back(j);
    }
    if (biases[i]) {
      PackJob j{};
      j.d.w = biases[i]; j.d.wp = p->bias; j.d.Ncol = d.cout;
      j.kind = 1;
      jobs.push_back(j);
    }
    if (fusable_second(p) && p->w2f) {
      PackJob j{};
      j.d.w = p->g[0].wp; j.d.wp = p->w2f; j.d.K = p->g[0].K; j.d.Ncol = p->g[0].Ncol;
      j.kind = 2;
      jobs2.push_back(j);
    }
  }
  std::vector<int> blk_job;
  std::vector<long long> blk_first;
  auto add_blocks = [&](const std::vector<PackJob>& js, int base) {
    for (size_t j = 0; j < js.size(); ++j) {
      const size_t total = js[j].kind == 0 ? (size_t)js[j].d.Ncol * js[j].d.K : js[j].kind == 1 ? (size_t)js[j].d.Ncol : (size_t)(2 * 3 * 12 * 64 * 4);
      for (size_t f = 0; f < total; f += kPackChunk) { blk_job.push_back(base + (int)j); blk_first.push_back((long long)f); }
    }
  };
  add_blocks(jobs, 0);
  const int nb1 = (int)blk_job.size();
  add_blocks(jobs2, (int)jobs.size());
  const int nb2 = (int)blk_job.size() - nb1;
  jobs.insert(jobs.end(), jobs2.begin(), jobs2.end());
  auto* g = new sntc_plan_group();
  g->nblocks = nb1;
  g->nblocks2 = nb2;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMalloc(&g->jobs, sizeof(PackJob) * jobs.size());
  if (e == hipSuccess) e = hipMalloc(&g->blk_job, sizeof(int) * blk_job.size());
  if (e == hipSuccess) e = hipMalloc(&g->blk_first, sizeof(long long) * blk_first.size());
  if (e == hipSuccess) e = hipMemcpyAsync(g->jobs, jobs.data(), sizeof(PackJob) * jobs.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(g->blk_job, blk_job.data(), sizeof(int) * blk_job.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(g->blk_first, blk_first.data(), sizeof(long long) * blk_first.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);      // the host vectors go out of scope
  if (e != hipSuccess) { sntc_plan_group_destroy(g); return hip_fail(e, "sntc_plan_group_create"); }
  *out = g;
  return SNTC_OK;
}

extern "C" int sntc_plan_group_update(const sntc_plan_group* g, void* stream) {
  if (!g) return fail(SNTC_ERR_BAD_SHAPE, "sntc_plan_group_update: null group");
  hipStream_t s = (hipStream_t)stream;
  if (g->nblocks > 0) hipLaunchKernelGGL(pack_group_kernel, dim3(g->nblocks), dim3(256), 0, s, g->jobs, g->blk_job, g->blk_first);
  if (g->nblocks2 > 0)
    hipLaunchKernelGGL(pack_group_kernel, dim3(g->nblocks2), dim3(256), 0, s, g->jobs, g->blk_job + g->nblocks, g->blk_first + g->nblocks);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

struct Geo {
  int Ho, Wo, Qh, Qw, sA, tstep, offy, offx, sO;
};

static int geometry(const sntc_conv_plan* p, int h, int w, Geo* g) {
  const sntc_conv_desc& d = p->d;
  const int s = d.stride;
  if (h < 1 || w < 1) return fail(SNTC_ERR_BAD_SHAPE, "empty image");
  if (p->rowpack) {                       // VALID on the caller-padded input (sntc_pad_zero): every tap of every output is in range
    if (h < d.kh || w < d.kw) return fail(SNTC_ERR_BAD_SHAPE, "row-packed plan: padded input smaller than the kernel");
    const int ho = (h - d.kh) / s + 1, wo = (w - d.kw) / s + 1;
    *g = Geo{ho, wo, ho, wo, s, 1, 0, 0, 1};
    return SNTC_OK;
  }
  if (!p->up) {
    g->Ho = (h + s - 1) / s;
    g->Wo = (w + s - 1) / s;
    int pt, pl;
    if (d.kind == SNTC_CONV2D) {
      pt = std::max((g->Ho - 1) * s + d.kh - h, 0) / 2;
      pl = std::max((g->Wo - 1) * s + d.kw - w, 0) / 2;
    } else {
      pt = p->pt;
      pl = p->pl;
    }
    *g = Geo{g->Ho, g->Wo, g->Ho, g->Wo, s, 1, -pt, -pl, 1};
  } else if (!p->phase_mode) {   // stride-1 transpose == correlation with the flipped kernel
    *g = Geo{h, w, h, w, 1, -1, p->pt, p->pl, 1};
  } else {
    const int ho = h * s, wo = w * s;
    if (p->exact_grid)      // each phase has exactly h x w valid macro pixels, offset by the group's origin
      *g = Geo{ho, wo, h, w, 1, -1, 0, 0, s};
    else
      *g = Geo{ho, wo, (ho - 1 + p->pt) / s + 1, (wo - 1 + p->pl) / s + 1, 1, -1, 0, 0, s};
  }
  return SNTC_OK;
}

extern "C" int sntc_conv_out_shape(const sntc_conv_plan* p, int h, int w, int* ho, int* wo) {
  if (!p || !ho || !wo) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_out_shape: null argument");
  Geo g;
  int rc = geometry(p, h, w, &g);
  if (rc) return rc;
  *ho = g.Ho;
  *wo = g.Wo;
  return SNTC_OK;
}

extern "C" int64_t sntc_conv_flops(const sntc_conv_plan* p, int n, int h, int w) {
  if (!p) return 0;
  Geo g;
  if (geometry(p, h, w, &g)) return 0;
  const sntc_conv_desc& d = p->d;
  const int64_t px = p->up ? (int64_t)h * w : (int64_t)g.Ho * g.Wo;
  return 2 * (int64_t)n * px * d.kh * d.kw * d.cin * d.cout;
}

// Launch schedule of one call: tile variant, split-K factor, and whether the persistent stream-K workers run it.
struct Sched {
  int variant = 2;
  int ksplit = 1;
  bool sk = false;
  int workers = 0;        // stream-K: resident workgroups
  int64_t units = 0;      // sum over groups of tiles * stages
  int64_t blocks = 0;     // workgroups launched
  bool deep = false;      // the deep-ring direct-to-LDS instance (small launches)
  bool valid = false;     // some candidate passed the filters (always true without a forced choice)
};

static int max_steps(const sntc_conv_plan* p) {
  int s = 0;
  for (int gi = 0; gi < p->ngroups; ++gi) s = std::max(s, p->g[gi].K / kStage);
  return s;
}

// Deterministic split-K factor: a function of the layer and the per-image geometry only (never of the
// batch size), so a given image gets bit-identical results alone or inside any batch.  Used where one
// image offers few output tiles but a long K loop (hyper transforms, SGA input-gradient convolutions).
static int pick_ksplit(const sntc_conv_plan* p, const Geo& g) {
  int steps_min = 1 << 30, steps_max = 0;
  int64_t bpi = 0;
  const int64_t mt = ((int64_t)g.Qh * g.Qw + 127) / 128;
  for (int gi = 0; gi < p->ngroups; ++gi) {
    const int steps = p->g[gi].K / 32;                 // in 32-deep units, as the round-1 rule was tuned
    steps_min = std::min(steps_min, steps);
    steps_max = std::max(steps_max, steps);
    bpi += mt * ((p->g[gi].Ncol + 63) / 64);
  }
#ifndef SNTC_KSPLIT_BPI_MAX
#define SNTC_KSPLIT_BPI_MAX 32
#endif
  if (steps_max < 64 || bpi > SNTC_KSPLIT_BPI_MAX) return 1;
  const int want = (int)((128 + bpi - 1) / bpi);
  return std::max(1, std::min({8, want, std::max(1, steps_min / 8)}));
}

static int variant_bm(int v) { return v > kNumVariants ? bf3p_variant_bm(v) : gg_variant_bm(v); }
static int variant_bn(int v) { return v > kNumVariants ? bf3p_variant_bn(v) : gg_variant_bn(v); }

static void count_work(const sntc_conv_plan* p, int v, int64_t M, int64_t* tiles, int64_t* units, double* padded_macs) {
  const int bm = variant_bm(v), bn = variant_bn(v);
  const int64_t ntm = (M + bm - 1) / bm;
  *tiles = 0; *units = 0; *padded_macs = 0;
  for (int gi = 0; gi < p->ngroups; ++gi) {
    const int64_t ntn = (p->g[gi].Ncol + bn - 1) / bn;
    *tiles += ntn * ntm;
    *units += ntn * ntm * (p->g[gi].K / kStage);
    *padded_macs += (double)p->g[gi].K * (double)(ntn * bn) * (double)(ntm * bm);
  }
}

// Tile variant: least padded multiply-adds x (rounds of resident workgroups actually run / rounds of work) -- the second
// factor is 1 under stream-K, where every worker gets the same number of stages -- divided by the measured relative MFMA
// rate of the tile shape.
constexpr int kDeepBlocksPerCU = 2;
constexpr double kDeepCost = 0.6;      // relative cost of a deep-ring launch against the rounds model below (measured, tools/b1_layers.py)

// Process-wide default of the stream-K schedule (sntc_conv_set_stream_k): on unless a caller has turned it off, e.g. after
// sntc_conv_status reported a timed-out hand-off on an oversubscribed device.  Results are bit-identical either way.
static std::atomic<int> g_stream_k_enabled{1};

// Pre-split bf16 x 3 plans: 256 x 256 or 256 x 128 tiles on one 512-thread workgroup per CU; stream-K whenever every CU gets
// at least a longest tile's worth of stages, else one workgroup per tile.  No split-K (layers that small stay on the fp32 path).
static Sched schedule_s3(const sntc_conv_plan* p, const Geo& g, int64_t n, const TuneChoice* force) {
  Sched best;
  const int64_t M = n * g.Qh * g.Qw;
  const int msteps = max_steps(p);
  const int cus = std::max(8, gg_num_cus());
  double best_cost = 1e300;
  for (int v : {11, 12, 13}) {
    if (p->tile >= 11 && v != p->tile) continue;
    if (force && v != force->variant) continue;
    int64_t tiles, units;
    double macs;
    count_work(p, v, M, &tiles, &units, &macs);
    Sched s;
    s.variant = v;
    s.ksplit = 1;
    s.units = units;
    const int64_t fit = msteps > 0 ? units / msteps : 0;
    const int workers = (int)(std::min<int64_t>(cus, fit) & ~7LL);
    // many short tiles (the 13x13 / 8 synthesis: 1482 tiles of 20 ... 80 stages on 256 CUs) balance by themselves and run
    // faster one workgroup per tile (160 vs 145 TFLOP/s-equivalent); few long ones need the stream-K cut (191 vs 155)
    const bool can_sk = g_stream_k_enabled.load(std::memory_order_relaxed) && units < (1LL << 31) && workers >= 8 && 2 * workers >= cus;
    s.sk = force ? (force->sk != 0 && can_sk) : (can_sk && !p->no_stream_k && (units >= 64 * tiles || p->force_stream_k));
    s.valid = true;
    s.workers = s.sk ? workers : 0;
    s.blocks = s.sk ? workers : tiles;
    double cost = macs;
    if (s.sk) {
      cost *= (double)cus / workers;
    } else {
      const double rounds = (double)tiles / cus;
      cost *= rounds < 1.0 ? 1.0 / rounds : std::ceil(rounds) / rounds;
    }
    // measured on the 480 -> 640 layer, padding aside: 256 x 128 (fragments double-buffered, bookkeeping inside the MFMA shadows)
    // 225 TFLOP/s-equivalent, 256 x 256 (single fragment set: 128 accumulators leave no room for a second) 245 on its padded tile
    cost /= (v == 11 ? 1.0 : v == 13 ? 0.97 : 0.92);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

static Sched schedule(const sntc_conv_plan* p, const Geo& g, int64_t n, bool fused = false, const TuneChoice* force = nullptr) {
  if (p->s3) return schedule_s3(p, g, n, force);
  Sched best;
  const int64_t M = n * g.Qh * g.Qw;
  const int ksplit = fused ? 1 : pick_ksplit(p, g);
  const int msteps = max_steps(p);
  const bool pro = p->d.prologue != SNTC_PRO_NONE;
  double best_cost = 1e300;
  for (int v = 1; v <= kNumVariants; ++v) {
    if (fused ? v != 3 : (p->tile >= 1 && p->tile <= kNumVariants && v != p->tile)) continue;
    // single-buffered fragments / one wave per SIMD: forced only.  (128 x 192 does beat 128 x 96 on the long N = 192
    // contractions when it has the device to itself -- 5x5 / 2, 192 -> 192: 130.9 vs 119.8 TFLOP/s -- but at two workgroups per
    // CU and 61 KB of LDS it shuts out the other stream's kernels: bench.py's two-stream encode went from 48.5 to 49.2 ms.)
    if (p->tile == 0 && (v == 6 || v == 7 || v == 10)) continue;
    if (p->bf3 && v != 2 && v != 4) continue;                       // the bf16 x 3 experiment is instantiated for two tile shapes
    if (force && v != force->variant) continue;
    int64_t tiles, units;
    double macs;
    count_work(p, v, M, &tiles, &units, &macs);
    const bool dma = !fused && plan_dma(p) && gg_resident_blocks_dma(v) > 0;
    const int resident = std::max(1, fused ? gg_resident_blocks_fused() : p->bf3 ? gg_resident_blocks_bf3(v) : dma ? gg_resident_blocks_dma(v) : gg_resident_blocks(v, p->vec, pro));
    Sched s;
    s.variant = v;
    s.ksplit = ksplit;
    s.units = units;
    // stream-K when every resident worker gets at least one longest tile's worth of stages (then a tile is shared by at
    // most two workers) and the unit count fits the kernel's 32-bit unit arithmetic
    // workers: every resident slot, or fewer when the launch is too small to give each of them a longest tile's worth
    // of stages (an idle slot costs less than a second, half-empty round of whole tiles); a multiple of 8 (XCD dealing)
    const int64_t fit = msteps > 0 ? units / msteps : 0;
    const int workers = (int)(std::min<int64_t>(resident, fit) & ~7LL);
    // ... and only where tiles are long: with fewer than 64 stages per tile on average (3x3 96 -> 96: 54, the 1x1 layers:
    // 6 ... 20, the 13x13 / 8 synthesis: 52) the pieces' own bookkeeping outweighs the tile quantisation it removes -- one
    // workgroup per tile measured +6 % on the 3x3 96 -> 96 layers, +4 ... 6 % on the 1x1 layers, +57 % on 320 -> 160 at 1/16
    // resolution, +4 % on the synthesis; long tiles (5x5 / 2: 300 stages, hyper-synthesis: 80 ... 270) keep stream-K (+6 ... 25 %)
    // ... and only where tiles are FEW: with four or more rounds of resident workgroups the hardware's own dispatch balances the
    // launch, and the hand-offs only cost (5x5/2 192 -> 192 at 256 x 384: 6912 tiles of 300 stages on 768 slots, 131.9 TFLOP/s one
    // workgroup per tile against 118.2 stream-K; at two rounds -- the decode layers -- stream-K wins by up to 25 %)
    const bool short_tiles = (units < 64 * tiles || tiles * ksplit >= 4 * (int64_t)resident) && !p->force_stream_k;
    const bool can_sk = ksplit == 1 && g_stream_k_enabled.load(std::memory_order_relaxed) && units < (1LL << 31) && workers >= 8 &&
                        2 * workers >= resident;
    s.sk = force ? (force->sk != 0 && can_sk) : (can_sk && !p->no_stream_k && !short_tiles);
    s.valid = true;
    s.workers = s.sk ? workers : 0;
    s.blocks = s.sk ? workers : tiles * ksplit;
    double cost = macs;
    // a launch of about one workgroup per CU or fewer has nothing but its own pipeline to hide the memory latency behind:
    // one image alone (Model.evaluate's reference flow), the hyper transforms.  Such a launch runs the deep-ring instance
    // (six stages in flight per workgroup instead of two) where the plan allows direct-to-LDS staging at all.  Same bits.
    s.deep = !fused && !s.sk && p->dma != 0 && p->vec && !p->rowpack && !pro && !p->bf3 && gg_resident_blocks_deep(v) > 0 &&
             tiles * ksplit <= (int64_t)kDeepBlocksPerCU * gg_num_cus();
    if (s.sk) {
      cost *= (double)resident / workers;
    } else if (s.deep) {
      cost *= std::max(1.0, (double)gg_num_cus() / (double)(tiles * ksplit)) * kDeepCost;
    } else {
      const double rounds = (double)(tiles * ksplit) / resident;
      cost *= rounds < 1.0 ? 1.0 / rounds : std::ceil(rounds) / rounds;
      if (tiles * ksplit < 768) cost *= 1.0 + 0.5 * (double)(768 - tiles * ksplit) / 768.0;
    }
    // measured MFMA rate of each tile shape on exact-fit layers, relative to 128 x 128 (tools/ab_layers.sh, round 2): a wave
    // that owns more accumulators reads fewer LDS bytes and stages fewer global bytes per MFMA, and the chip holds a
    // higher clock for it; 256 x 128 leaves one wave per SIMD and stalls on every barrier
    // (round 3, back-to-back launches: 128 x 96 beats 128 x 128 by 3.5 % on the 320 -> 480 layer whose 480 columns it fits exactly,
    // where the old 0.97 let the stream-K worker count tip the choice the other way)
    static const double kRate[kNumVariants + 1] = {0, 0.86, 0.93, 1.00, 1.00, 0.97, 0.90, 0.85, 0.90, 1.02, 0.40};
    cost /= kRate[v];
#ifdef SNTC_DIAG
    if (getenv("SNTC_SCHED_DBG"))
      fprintf(stderr, "[sched] M=%lld v=%d sk=%d workers=%d blocks=%lld tiles=%lld units=%lld resident=%d deep=%d cost=%.4g (macs %.4g)\n", (long long)M, v,
              (int)s.sk, s.workers, (long long)s.blocks, (long long)tiles, (long long)units, resident, (int)s.deep, cost, macs);
#endif
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

// The schedule of a call: the measured choice for this (n, h, w) if sntc_conv_plan_tune recorded one -- unless the plan's tile or
// schedule is forced (tests, profiling) -- else the cost model's.
static Sched plan_schedule(const sntc_conv_plan* p, const Geo& g, int n, int h, int w, bool fused = false) {
  if (!fused && p->tile == 0 && !p->no_stream_k && !p->force_stream_k) {
    TuneChoice c;
    bool have = false;
    {
      std::lock_guard<std::mutex> lk(p->tune_mu);
      auto it = p->tuned.find({n, h, w});
      if (it != p->tuned.end()) { c = it->second; have = true; }
    }
    // a measured stream-K choice says nothing about the static schedules: with stream-K switched off process-wide
    // (sntc_conv_set_stream_k(0): a timed-out hand-off, or launches that run beside long-lived kernels) the cost model picks
    // among the static candidates instead of running the stream-K tile one workgroup per tile
    if (have && c.sk != 0 && !g_stream_k_enabled.load(std::memory_order_relaxed)) have = false;
    if (have) {
      const Sched s = schedule(p, g, n, false, &c);
      if (s.valid) return s;
    }
  }
  return schedule(p, g, n, fused);
}

static int64_t workspace_floats(const sntc_conv_plan* p, int64_t M, const Sched& s) {
  if (s.sk)      // slabs + one flag per worker
    return (int64_t)s.workers * (int64_t)(s.variant > kNumVariants ? bf3p_sk_slab_floats(s.variant) : gg_sk_slab_floats(s.variant)) + s.workers;
  if (s.ksplit <= 1) return 0;
  int64_t cols = 0;
  for (int gi = 0; gi < p->ngroups; ++gi) cols += p->g[gi].Ncol;
  return (int64_t)s.ksplit * M * cols;
}

extern "C" int64_t sntc_conv_workspace_bytes(const sntc_conv_plan* p, int n, int h, int w) {
  if (!p) return 0;
  Geo g;
  if (geometry(p, h, w, &g)) return 0;
  return 4 * workspace_floats(p, (int64_t)n * g.Qh * g.Qw, plan_schedule(p, g, n, h, w));
}

// fp32 stream-K unit order of a launch: column tile outermost for single-group plans whose packed weights do not fit an XCD's
// 4 MB L2, where the twin of the kernel exists (csrc/gather_gemm.hip, COLM)
static bool column_major(const sntc_conv_plan* p, const Sched& sc, int dma) {
  return !p->s3 && !p->bf3 && sc.sk && p->ngroups == 1 && p->colm != 0 && gg_colm_available(sc.variant, p->vec, p->d.prologue, dma) &&
         (p->colm == 1 || (size_t)p->g[0].Ncol * p->g[0].K * sizeof(float) > ((size_t)4 << 20));
}

extern "C" int sntc_conv_launch_order(const sntc_conv_plan* p, int n, int h, int w, int* column_major_out) {
  if (!p || !column_major_out) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_launch_order: null argument");
  Geo g;
  int rc = geometry(p, h, w, &g);
  if (rc) return rc;
  const Sched s = plan_schedule(p, g, n, h, w);
  *column_major_out = column_major(p, s, s.deep ? 2 : plan_dma(p) ? 1 : 0) ? 1 : 0;
  return SNTC_OK;
}

extern "C" int sntc_conv_launch_info(const sntc_conv_plan* p, int n, int h, int w, int* variant, int* nblocks) {
  if (!p || !variant || !nblocks) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_launch_info: null argument");
  Geo g;
  int rc = geometry(p, h, w, &g);
  if (rc) return rc;
  const Sched s = plan_schedule(p, g, n, h, w);
  *variant = s.variant;
  *nblocks = (int)s.blocks;
  return SNTC_OK;
}

// One launch of plan p; with p2 (validated by sntc_conv_forward_fused) the 1x1 plan p2 runs behind p inside the same launch.
static int conv_forward_impl(const sntc_conv_plan* p, const sntc_conv_plan* p2, const float* x, int n, int h, int w, float* y,
                             const float* res, const float* aux, void* workspace, size_t workspace_bytes, void* stream,
                             const TuneChoice* force = nullptr) {
  if (!p || !x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_forward: null argument");
  if (n < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_forward: empty batch");
  const sntc_conv_desc& d = p->d;
  const int epilogue = p2 ? p2->d.epilogue : d.epilogue;
  if (epilogue != SNTC_EPI_STORE && !res) return fail(SNTC_ERR_BAD_SHAPE, "epilogue needs `res`");
  if (epilogue == SNTC_EPI_GATE && !aux) return fail(SNTC_ERR_BAD_SHAPE, "gate epilogue needs `aux`");
  Geo g;
  int rc = geometry(p, h, w, &g);
  if (rc) return rc;
  const int64_t M = (int64_t)n * g.Qh * g.Qw;
  const int64_t x_bytes = (int64_t)n * h * w * d.cin * (p->s3 ? 6 : 4);
  if (M > 0x7fffffffLL || x_bytes >= (1LL << 31))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_forward: input tensor must be < 2 GiB (32-bit buffer offsets); split the batch");
  if (p2 && M * p2->d.cout >= (1LL << 32)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_forward_fused: output too large; split the batch");
  const Sched sc = force ? schedule(p, g, n, false, force) : plan_schedule(p, g, n, h, w, p2 != nullptr);
  if (!sc.valid) return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_forward: no tile variant for this plan");
  const int64_t ws_floats = workspace_floats(p, M, sc);
  if (ws_floats > 0 && (!workspace || workspace_bytes < (size_t)ws_floats * 4))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_forward: this call needs sntc_conv_workspace_bytes() of workspace "
                                    "(split-K slabs / stream-K hand-off)");
  const int v = sc.variant;
  const int bm = variant_bm(v), bn = variant_bn(v);
  GGArgs a{};
  a.ksplit = sc.ksplit;
  a.slab = static_cast<float*>(workspace);
  a.x = x; a.y = y; a.bias = p->bias; a.res = res; a.aux = aux;
  a.x_bytes = (unsigned)x_bytes;
  a.N = n; a.H = h; a.W = w; a.Cin = d.cin;
  a.Qh = g.Qh; a.Qw = g.Qw; a.M = (int)M;
  a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = d.cout;
  a.sA = g.sA; a.tstep = g.tstep; a.offy = g.offy; a.offx = g.offx; a.sO = g.sO;
  a.act = d.act; a.epi = epilogue; a.pro = d.prologue;
  a.ntm = (int)((M + bm - 1) / bm);
  a.ngroups = p->ngroups;
  a.bf3 = p->bf3 ? 1 : 0;
  a.dma = p2 ? 0 : sc.deep ? 2 : plan_dma(p) ? 1 : 0;
  if (p2) { a.w2f = p2->w2f; a.bias2 = p2->bias; a.Cout2 = p2->d.cout; }
  a.status = gg_status_word();
  if (!a.status) return fail(SNTC_ERR_HIP, "sntc_conv_forward: the device tables of the current device are not initialised");
  a.sk = sc.sk ? 1 : 0;
  a.nworkers = sc.workers;
  a.units = sc.units;
  if (sc.sk) {
    a.sk_slab = static_cast<float*>(workspace);
    a.sk_flags = reinterpret_cast<int*>(a.sk_slab + (size_t)sc.workers * (v > kNumVariants ? bf3p_sk_slab_floats(v) : gg_sk_slab_floats(v)));
    a.slab = nullptr;
    if (int zrc = zero_async(a.sk_flags, sizeof(int) * sc.workers, (hipStream_t)stream)) return zrc;
  }
  int nb = 0, tile0 = 0;
  long long unit0 = 0;
  size_t slab_off = 0;
  for (int gi = 0; gi < p->ngroups; ++gi) {
    GGGroup& G = a.g[gi];
    G.wp = p->g[gi].wp; G.taps = p->g[gi].taps; G.cols = reinterpret_cast<const int*>(p->g[gi].cols);
    G.T = p->g[gi].T; G.K = p->g[gi].K; G.Ncol = p->g[gi].Ncol; G.tw = p->g[gi].tw;
    G.ntn = (G.Ncol + bn - 1) / bn;
    G.steps = G.K / kStage;
    G.blk0 = nb;
    G.tile0 = tile0;
    G.unit0 = unit0;
    nb += G.ntn * a.ntm * sc.ksplit;
    tile0 += G.ntn;
    unit0 += (long long)G.ntn * G.steps;
    G.slab_off = slab_off;
    G.q0y = p->g[gi].q0y; G.q0x = p->g[gi].q0x;
    slab_off += (size_t)sc.ksplit * M * G.Ncol;
  }
  a.tps = tile0;
  a.ups = (int)unit0;
  if (p->s3) {
    // stream-K unit order: strip-major, as the fp32 kernel (every worker's share mixes the phase groups; with the column tile
    // outermost the last workers of the 13x13/8 synthesis get nothing but 20-stage tiles: 0.55 against 0.45 ms).  The column-major
    // order stays selectable for the A/B: sntc_conv_plan_set_schedule's stage-path bit ("off") doubles as the switch here
    a.order = p->dma == 0 ? 0 : 1;
    // patch staging: the taps of a slab sample the input at unit stride, and every group's patch (tile rows + the tap window's
    // reach in flattened macro pixels + a zero row) fits the patch buffers; <= 32 taps (the kernel keeps one validity bit per tap
    // and row)
    bool halo = !p->no_halo && g.sA == 1;
    for (int gi = 0; halo && gi < p->ngroups; ++gi) {
      const int th = p->g[gi].T / p->g[gi].tw;
      halo = p->g[gi].T <= 32 && bm + (th - 1) * g.Qw + p->g[gi].tw - 1 + (p->g[gi].T > 1 ? 1 : 0) <= bf3p_patch_rows_max();
    }
    a.halo = halo ? 1 : 0;
    return bf3p_launch(v, a, sc.sk ? sc.workers : nb, (hipStream_t)stream);
  }
  // stream-K unit order of the fp32 kernel: column tile outermost (GGArgs::order == 0) for single-group plans whose packed
  // weights do not fit an XCD's 4 MB L2 -- the 3x3 hyper-synthesis layer (11 MB): HBM-side reads of the launch 1667 -> 857 MB
  a.order = (!p2 && column_major(p, sc, a.dma)) ? 0 : 1;
  rc = gg_launch(v, p->vec, a, sc.sk ? sc.workers : nb, (hipStream_t)stream);
  if (rc || sc.sk || sc.ksplit <= 1) return rc;
  return gg_reduce_launch(a, (hipStream_t)stream);
}

extern "C" int sntc_conv_forward(const sntc_conv_plan* p, const float* x, int n, int h, int w, float* y,
                                 const float* res, const float* aux, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return conv_forward_impl(p, nullptr, x, n, h, w, y, res, aux, workspace, workspace_bytes, stream);
}

// ---- measured schedule (reference: none -- TensorFlow picks its convolution algorithm by cuDNN autotune the same way).
// Every (tile variant, schedule) candidate of a plan computes the same k-ordered fma chains (tests/test_hip_fullsize.py), so which
// one runs is a question of speed only.  The cost model of schedule() ranks them from tile counts; this measures them on the
// caller's buffers, for one (n, h, w), and records the winner in the plan.  Split-K factors are not candidates (they are a
// function of the layer and the per-image geometry only, by contract).
static void tune_candidates(const sntc_conv_plan* p, const Geo& g, int n, std::vector<std::pair<TuneChoice, Sched>>* out) {
  const int v0 = p->s3 ? 11 : 1, v1 = p->s3 ? 13 : kNumVariants;
  for (int v = v0; v <= v1; ++v)
    for (int sk = 1; sk >= 0; --sk) {
      TuneChoice c;
      c.variant = v;
      c.sk = sk;
      const Sched s = schedule(p, g, n, false, &c);
      if (!s.valid || s.variant != v || (sk && !s.sk)) continue;
      out->push_back({c, s});
    }
}

extern "C" int64_t sntc_conv_tune_workspace_bytes(const sntc_conv_plan* p, int n, int h, int w) {
  if (!p) return 0;
  Geo g;
  if (geometry(p, h, w, &g)) return 0;
  std::vector<std::pair<TuneChoice, Sched>> cand;
  tune_candidates(p, g, n, &cand);
  int64_t m = 0;
  for (auto& c : cand) m = std::max(m, workspace_floats(p, (int64_t)n * g.Qh * g.Qw, c.second));
  return 4 * m;
}

extern "C" int sntc_conv_plan_tune(sntc_conv_plan* p, const float* x, int n, int h, int w, float* y, const float* res,
                                   const float* aux, void* workspace, size_t workspace_bytes, int reps, int* variant,
                                   int* stream_k, void* stream) {
  if (!p || !x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_tune: null argument");
  Geo g;
  int rc = geometry(p, h, w, &g);
  if (rc) return rc;
  reps = std::max(1, std::min(reps, 100));
  std::vector<std::pair<TuneChoice, Sched>> cand;
  tune_candidates(p, g, n, &cand);
  if (cand.empty()) return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_plan_tune: no candidate");
  hipEvent_t e0, e1;
  SNTC_HIP(hipEventCreate(&e0));
  SNTC_HIP(hipEventCreate(&e1));
  float best_ms = 1e30f;
  TuneChoice best;
  for (auto& c : cand) {
    if ((size_t)(4 * workspace_floats(p, (int64_t)n * g.Qh * g.Qw, c.second)) > workspace_bytes) continue;
    rc = conv_forward_impl(p, nullptr, x, n, h, w, y, res, aux, workspace, workspace_bytes, stream, &c.first);      // warm
    if (rc) break;
    (void)hipEventRecord(e0, (hipStream_t)stream);
    for (int r = 0; r < reps && !rc; ++r)
      rc = conv_forward_impl(p, nullptr, x, n, h, w, y, res, aux, workspace, workspace_bytes, stream, &c.first);
    (void)hipEventRecord(e1, (hipStream_t)stream);
    if (rc) break;
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = fail(SNTC_ERR_HIP, "sntc_conv_plan_tune: timing failed"); break; }
    if (ms < best_ms) { best_ms = ms; best = c.first; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  if (best.variant == 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_tune: workspace smaller than sntc_conv_tune_workspace_bytes()");
  {
    std::lock_guard<std::mutex> lk(p->tune_mu);
    p->tuned[{n, h, w}] = best;
  }
  if (variant) *variant = best.variant;
  if (stream_k) *stream_k = best.sk;
  // leave y as a plain forward call would: the last candidate's output is the same bits, nothing to redo
  return SNTC_OK;
}

extern "C" int sntc_conv_plan_set_choice(sntc_conv_plan* p, int n, int h, int w, int variant, int stream_k) {
  if (!p) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_set_choice: null plan");
  Geo g;
  int rc = geometry(p, h, w, &g);
  if (rc) return rc;
  TuneChoice c;
  c.variant = variant;
  c.sk = stream_k ? 1 : 0;
  const Sched s = schedule(p, g, n, false, &c);
  if (!s.valid || s.variant != variant || (stream_k && !s.sk))
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_plan_set_choice: not a candidate of this plan for this call shape");
  std::lock_guard<std::mutex> lk(p->tune_mu);
  p->tuned[{n, h, w}] = c;
  return SNTC_OK;
}

extern "C" int sntc_conv_plan_candidates(const sntc_conv_plan* p, int n, int h, int w, int* variants, int* stream_k, int capacity) {
  if (!p || !variants || !stream_k || capacity < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_candidates: bad argument");
  Geo g;
  if (geometry(p, h, w, &g)) return -1;
  std::vector<std::pair<TuneChoice, Sched>> cand;
  tune_candidates(p, g, n, &cand);
  int k = 0;
  for (auto& c : cand)
    if (k < capacity) { variants[k] = c.first.variant; stream_k[k] = c.first.sk; ++k; }
  return k;
}

extern "C" int sntc_conv_plan_clear_tuning(sntc_conv_plan* p) {
  if (!p) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_plan_clear_tuning: null plan");
  std::lock_guard<std::mutex> lk(p->tune_mu);
  p->tuned.clear();
  return SNTC_OK;
}

// ResidualBlock tail (reference common/elic.py:41-68): y = epilogue2(conv1x1_p2(act1(conv3x3_p1(x) + b1)) + b2, res) in ONE
// launch -- the 96-channel intermediate never leaves the registers.  Bit-identical to the two launches.
static const char* fused_pair_error(const sntc_conv_plan* p, const sntc_conv_plan* p2) {
  if (!p || !p2) return "null plan";
  const sntc_conv_desc& a = p->d;
  if (p->up || p2->up || p->bf3 || p2->bf3) return "forward convolutions in fp32 only";
  if (!p->vec || a.cout != 96 || a.stride != 1 || a.prologue != SNTC_PRO_NONE || a.epilogue != SNTC_EPI_STORE || p->ngroups != 1)
    return "first plan: stride-1 convolution with Cin % 16 == 0 and 96 output channels, plain store";
  if (!fusable_second(p2) || !p2->w2f) return "second plan: 1x1 convolution 96 -> 192 without activation";
  return nullptr;
}

extern "C" int sntc_conv_fusable(const sntc_conv_plan* p, const sntc_conv_plan* p2) { return fused_pair_error(p, p2) ? 0 : 1; }

extern "C" int64_t sntc_conv_fused_workspace_bytes(const sntc_conv_plan* p, int n, int h, int w) {
  if (!p) return 0;
  Geo g;
  if (geometry(p, h, w, &g)) return 0;
  return 4 * workspace_floats(p, (int64_t)n * g.Qh * g.Qw, schedule(p, g, n, true));
}

extern "C" int sntc_conv_forward_fused(const sntc_conv_plan* p, const sntc_conv_plan* p2, const float* x, int n, int h, int w, float* y,
                                       const float* res, const float* aux, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  if (const char* why = fused_pair_error(p, p2)) return fail(SNTC_ERR_UNSUPPORTED, why);
  return conv_forward_impl(p, p2, x, n, h, w, y, res, aux, workspace, workspace_bytes, stream);
}

// ---- stream-K health: a worker that waits in vain for its neighbour's hand-off flags the launch instead of
// trapping; the host asks at a point where it synchronises anyway.
extern "C" int sntc_conv_set_stream_k(int enabled) {
  g_stream_k_enabled.store(enabled ? 1 : 0, std::memory_order_relaxed);
  return SNTC_OK;
}

extern "C" int sntc_conv_get_stream_k(void) { return g_stream_k_enabled.load(std::memory_order_relaxed); }

// read-and-clear in ONE atomic: a flag raised by a launch on another stream between a copy and a later memset would be lost
__global__ void status_exchange_kernel(int* word, int* out) { *out = atomicExch(word, 0); }
__global__ void status_or_kernel(int* word, int flags) { atomicOr(word, flags); }

extern "C" int sntc_conv_status(int* flags, void* stream) {
  if (!flags) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_status: null argument");
  int rc = gg_init();
  if (rc) return rc;
  int* word = gg_status_word();      // two ints: the sticky word, and the slot the exchange below returns it through
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(status_exchange_kernel, dim3(1), dim3(1), 0, s, word, word + 1);
  SNTC_HIP(hipGetLastError());
  SNTC_HIP(hipMemcpyAsync(flags, word + 1, sizeof(int), hipMemcpyDeviceToHost, s));
  SNTC_HIP(hipStreamSynchronize(s));
  return SNTC_OK;
}

extern "C" int sntc_conv_status_inject(int flags, void* stream) {
  int rc = gg_init();
  if (rc) return rc;
  hipLaunchKernelGGL(status_or_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gg_status_word(), flags);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

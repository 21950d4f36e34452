// wgrad.hip -- weight gradients of Conv2D / Conv2DTranspose (Model.train_step,
// mshyper/models.py:375-383 -> tape.gradient w.r.t. every kernel) on the fp32 matrix cores.
//
// Both layer kinds reduce to ONE contraction over the pixels of the layer's LOW-resolution side:
//     dW[tap][a][b] = sum_{n,i,j}  S[n, i*s + ky - pt, j*s + kx - pl, a] * D[n, i, j, b]        (tap = ky*kw + kx)
//   Conv2D          (kernel [kh,kw,Cin,Cout]):  S = layer input x,            D = grad of the pre-activation output
//   Conv2DTranspose (kernel [kh,kw,Cout,Cin]):  S = grad of the pre-act output, D = layer input x
// i.e. a GEMM with M = taps*Cs (m = tap*Cs + a: the gather dimension, as K is in the forward kernel),
// N = Cd and K = n*Hd*Wd pixels.  K is cut into slabs over blocks (split-K: the outputs are small, the reduction
// is huge); each block writes its partial tile to a workspace slab and a second kernel sums the slabs in a fixed
// order (deterministic), optionally accumulating into dW.
//
// Block = 4 waves covering 128 x (32..192) outputs, MFMA v_mfma_f32_32x32x2_f32 (exact f32).  The operands arrive
// pixel-major ([pixel][channel], NHWC rows), which is exactly [k][m] / [k][n]: the LDS tiles keep that layout and
// every MFMA operand is one ds_read_b32 per lane (lanes 0-31: 32 consecutive channels of pixel k, lanes 32-63 of
// pixel k + 1) -- conflict-free without padding.  Global loads are 16 B per lane along channels, zero outside the
// image (padding taps) or beyond M / N / the slab.
#include <algorithm>
#include <cstdlib>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WGArgs {
  const float* S;      // [n, Hs, Ws, Cs]
  const float* D;      // [n, Hd, Wd, Cd]
  float* slab;         // [ksplit][M][N] partial sums (ksplit > 1) or dW itself (ksplit == 1)
  int accumulate;      // ksplit == 1 only: add to the existing dW
  int n, Hs, Ws, Cs, Hd, Wd, Cd;
  int kh, kw, stride, pt, pl;
  int M, N;            // taps*Cs, Cd
  long long P;         // n*Hd*Wd
  long long pslab;     // pixels per K slab (multiple of BK)
  int ntm, ntn, ksplit;
  int dbg;             // diagnostic switches (SNTC_WG_DBG): 1 skip global loads, 2 skip LDS writes, 4 skip barriers
};

constexpr int kBK = 16;     // pixels per LDS stage (32 measured slower: fewer, fatter K slabs and half the blocks per CU)
constexpr int kAPass = kBK / 8;   // A-tile rows per thread per stage (256 threads = 8 rows x 32 float4)
constexpr int kBM = 128;

// WN waves across N (1 or 2), each TN MFMA tiles wide; the other 4 / WN waves stack along M with TM tiles each so
// that the block always covers 128 rows: BN = 32 WN TN in {32, 64, 96, 128, 160, 192} fits N = 96 / 160 / 192 / 320 /
// 480 exactly (no MFMA work on padding columns).
template <int WN, int TN, bool VECS>
__global__ void __launch_bounds__(256) wgrad_kernel(WGArgs a) {
  constexpr int WM = 4 / WN, TM = kBM / (32 * WM);
  constexpr int BN = 32 * WN * TN;
  __shared__ float As[2][kBK][kBM];
  __shared__ float Bs[2][kBK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  const int tm = bid % a.ntm;
  bid /= a.ntm;
  const int tn = bid % a.ntn;
  const int ks = bid / a.ntn;
  const int m0 = tm * kBM, n0 = tn * BN;
  const long long p0 = (long long)ks * a.pslab, p1 = std::min(a.P, p0 + a.pslab);

  // ---- loader roles: A tile = 16 rows x 32 float4 -> 2 per thread; B tile = 16 rows x (BN / 4) float4
  const int ac4 = tid & 31, ar0 = tid >> 5;                 // rows ar0, ar0 + 8
  const int am = m0 + 4 * ac4;
  int aky[4], akx[4], acs[4];
  bool aok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = am + e;
    aok[e] = m < a.M;
    const int tap = aok[e] ? m / a.Cs : 0;
    acs[e] = aok[e] ? m - tap * a.Cs : 0;
    aky[e] = tap / a.kw - a.pt;
    akx[e] = tap % a.kw - a.pl;
  }
  constexpr int BC4 = BN / 4;                               // float4 columns of the B tile
  constexpr int BTOT = kBK * BC4;                           // float4 per stage
  constexpr int BPASS = (BTOT + 255) / 256;

  // The pixel rows this thread gathers per stage, as (i, j) + the element offset of S[img, i*s, j*s, 0]: decoded once,
  // then advanced by kBK pixels per stage (no division, 32-bit offsets: tensors are < 2^31 elements, host check).
  int pi[kAPass], pj[kAPass];
  unsigned sbase[kAPass];
  const unsigned step_j = (unsigned)(a.stride * a.Cs), step_i = (unsigned)(a.stride * a.Ws * a.Cs);
  const unsigned wrap_j = (unsigned)a.Wd * step_j;                       // what a full row of D pixels adds along j
  const unsigned next_img = (unsigned)(a.Hs * a.Ws * a.Cs) - (unsigned)a.Hd * step_i;   // (img + 1, 0) - (img, Hd)
#pragma unroll
  for (int q = 0; q < kAPass; ++q) {
    const long long p = p0 + ar0 + 8 * q;
    const long long hw = (long long)a.Hd * a.Wd;
    const int img = (int)(p / hw);
    const int rem = (int)(p - (long long)img * hw);
    pi[q] = rem / a.Wd;
    pj[q] = rem - pi[q] * a.Wd;
    sbase[q] = (unsigned)img * (unsigned)(a.Hs * a.Ws * a.Cs) + (unsigned)pi[q] * step_i + (unsigned)pj[q] * step_j;
  }
  // tap offsets of this thread's four gathered columns, relative to S[img, i*s, j*s, 0]
  int toff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) toff[e] = (aky[e] * a.Ws + akx[e]) * a.Cs + acs[e];
  unsigned boff[BPASS];                                                  // element offsets into D of this thread's B loads
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx / BC4, c4 = idx - row * BC4;
    boff[q] = (unsigned)(p0 + row) * (unsigned)a.Cd + (unsigned)(n0 + 4 * c4);
  }
  f32x4 ra[kAPass], rb[BPASS];
  auto gload = [&](long long pb) {
    const int left = (int)std::min<long long>(p1 - pb, kBK);             // valid pixel rows in this stage
#pragma unroll
    for (int q = 0; q < kAPass; ++q) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ar0 + 8 * q < left) {
        const int i = pi[q], j = pj[q];
        if (VECS) {
          const int sy = i * a.stride + aky[0], sx = j * a.stride + akx[0];
          if (aok[0] && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws)
            v = *reinterpret_cast<const f32x4*>(a.S + (int)(sbase[q] + (unsigned)toff[0]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int sy = i * a.stride + aky[e], sx = j * a.stride + akx[e];
            if (aok[e] && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws) v[e] = a.S[(int)(sbase[q] + (unsigned)toff[e])];
          }
        }
      }
      ra[q] = v;
      pj[q] += kBK;                                         // next stage: kBK pixels further
      sbase[q] += kBK * step_j;
      while (pj[q] >= a.Wd) {
        pj[q] -= a.Wd;
        sbase[q] += step_i - wrap_j;
        if (++pi[q] == a.Hd) { pi[q] = 0; sbase[q] += next_img; }
      }
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / BC4, c4 = idx - row * BC4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < BTOT && row < left && n0 + 4 * c4 < a.N) v = *reinterpret_cast<const f32x4*>(a.D + boff[q]);   // N % 4 == 0
      rb[q] = v;
      boff[q] += (unsigned)(kBK * a.Cd);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < kAPass; ++q) *reinterpret_cast<f32x4*>(&As[buf][ar0 + 8 * q][4 * ac4]) = ra[q];
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / BC4, c4 = idx - row * BC4;
      if (idx < BTOT) *reinterpret_cast<f32x4*>(&Bs[buf][row][4 * c4]) = rb[q];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kl = lane >> 5, cl = lane & 31;
  const int arow = wm * 32 * TM + cl, bcol = wn * 32 * TN + cl;
  int buf = 0;
  if (p0 < p1) {
    gload(p0);
    lstore(0);
  }
  __syncthreads();
  for (long long pb = p0; pb < p1; pb += kBK) {
    const bool more = pb + kBK < p1;
    if (more && !SNTC_DBG(a, 1)) gload(pb + kBK);
#pragma unroll
    for (int kk = 0; kk < kBK / 2; ++kk) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = As[buf][2 * kk + kl][arow + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = Bs[buf][2 * kk + kl][bcol + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more && !SNTC_DBG(a, 2)) lstore(buf ^ 1);
    if (!SNTC_DBG(a, 4)) __syncthreads();
    buf ^= 1;
  }

  float* out = a.slab + (size_t)ks * a.M * a.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + cl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * TM + 32 * i + (r >> 2) * 8 + kl * 4 + (r & 3);
        if (row < a.M && col < a.N) {
          float* o = out + (size_t)row * a.N + col;
          *o = (a.ksplit == 1 && a.accumulate) ? *o + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

// dW[e] (= or +=) sum over slabs in a FIXED association order (deterministic): a block owns 64 elements, its four
// waves take the slabs k = wave, wave + 4, ... (four loads in flight each), combined through LDS
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, long long total, int ksplit,
                                                           float* __restrict__ dw, int accumulate) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long e = blockIdx.x * 64LL + l;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < total) {
    int k = grp;
    for (; k + 12 < ksplit; k += 16) {
      s0 += slab[(size_t)k * total + e];
      s1 += slab[(size_t)(k + 4) * total + e];
      s2 += slab[(size_t)(k + 8) * total + e];
      s3 += slab[(size_t)(k + 12) * total + e];
    }
    for (; k < ksplit; k += 4) s0 += slab[(size_t)k * total + e];
  }
  red[grp][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < total) {
    const float s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    dw[e] = accumulate ? dw[e] + s : s;
  }
}

// db[c] (= or +=) sum over pixels of g[p][c].  HBM-bound: a block owns 64 channels x one pixel slab, a thread one
// float4 of channels and every 16th pixel (4 loads in flight), LDS-reduced to one partial row per slab; the partial rows
// are summed in a fixed order by wgrad_reduce_kernel.
template <bool VEC>
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ g, long long P, int C, long long pslab,
                                                     float* __restrict__ part) {
  __shared__ f32x4 red[16][16];
  const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * q;
  const long long p0 = blockIdx.y * pslab, p1 = std::min(P, p0 + pslab);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  auto ld = [&](long long p) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p < p1 && c < C) {
      if (VEC) v = *reinterpret_cast<const f32x4*>(g + (size_t)p * C + c);
      else
        for (int e = 0; e < 4; ++e)
          if (c + e < C) v[e] = g[(size_t)p * C + c + e];
    }
    return v;
  };
  for (long long p = p0 + r; p < p1; p += 64) {
    const f32x4 a0 = ld(p), a1 = ld(p + 16), a2 = ld(p + 32), a3 = ld(p + 48);
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  red[r][q] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (r == 0) {
    f32x4 t = red[0][q];
    for (int i = 1; i < 16; ++i) t += red[i][q];
    for (int e = 0; e < 4; ++e)
      if (c + e < C) part[(size_t)blockIdx.y * C + c + e] = t[e];
  }
}

}  // namespace sntc

using namespace sntc;

namespace {

struct WGGeo {
  int Hs, Ws, Cs, Hd, Wd, Cd, pt, pl;
  bool x_is_S;
};

// kind: SNTC_CONV2D / SNTC_CONV2D_TRANSPOSE / SNTC_SIGNAL_DOWN / SNTC_SIGNAL_UP; (h, w) = spatial size of the layer INPUT x
int wg_geometry(int kind, int kh, int kw, int stride, int cin, int cout, int h, int w, WGGeo* g) {
  if (kh < 1 || kw < 1 || stride < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_wgrad: bad sizes");
  if (kind == SNTC_CONV2D) {                       // Keras SAME: out = ceil(in / s), pad_before = pad_total / 2
    const int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
    *g = WGGeo{h, w, cin, ho, wo, cout, std::max((ho - 1) * stride + kh - h, 0) / 2, std::max((wo - 1) * stride + kw - w, 0) / 2, true};
  } else if (kind == SNTC_CONV2D_TRANSPOSE) {      // Keras SAME: out = in * s, pad_before = max(k - s, 0) / 2
    *g = WGGeo{h * stride, w * stride, cout, h, w, cin, std::max(kh - stride, 0) / 2, std::max(kw - stride, 0) / 2, false};
  } else if (kind == SNTC_SIGNAL_DOWN) {           // tfc same_zeros, corr=True: centred kernel, pad_before = k / 2
    const int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
    *g = WGGeo{h, w, cin, ho, wo, cout, kh / 2, kw / 2, true};
  } else if (kind == SNTC_SIGNAL_UP) {             // tfc same_zeros, corr=False, strides_up: out[s i + j - (k-1)/2] += x[i] w[j]
    // dW comes out as [kh, kw, Cout, Cin] (high-resolution side first), i.e. channel-transposed with respect to the
    // SignalConv2D kernel [kh, kw, Cin, Cout]: the caller transposes the last two axes (sntc_transpose_last2)
    *g = WGGeo{h * stride, w * stride, cout, h, w, cin, (kh - 1) / 2, (kw - 1) / 2, false};
  } else {
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_wgrad: unknown layer kind");
  }
  if (g->Cd % 4) return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_wgrad: the low-resolution side needs a channel count divisible by 4");
  return SNTC_OK;
}

// column-tile width: the candidate with the least padded columns (ties: the wider one, more reuse of the A tile)
int wg_pick_bn(int N) {
  static const int cand[6] = {32, 64, 96, 128, 160, 192};
  int best = 32;
  long long best_pad = 1LL << 60;
  for (int c : cand) {
    const long long padded = (long long)((N + c - 1) / c) * c;
    if (padded <= best_pad) { best_pad = padded; best = c; }
  }
  return best;
}

void wg_split(const WGGeo& g, int kh, int kw, int n, int* bn_out, int* ntm, int* ntn, int* ksplit, long long* pslab) {
  const int M = kh * kw * g.Cs, N = g.Cd;
  const int bn = wg_pick_bn(N);
  *bn_out = bn;
  *ntm = (M + kBM - 1) / kBM;
  *ntn = (N + bn - 1) / bn;
  const long long P = (long long)n * g.Hd * g.Wd;
  const long long tiles = (long long)*ntm * *ntn;
  const long long chunks = (P + kBK - 1) / kBK;
  long long ks = std::max<long long>(1, (1536 + tiles - 1) / tiles);       // ~6 blocks per CU in flight
  // every slab costs a write + a read of the whole [M, N] output: keep >= 24 stages (384 pixels) of MFMA work per
  // slab so that this traffic stays a fraction of the contraction (pixel-poor, weight-heavy hyper layers: no split)
  ks = std::min(ks, std::max<long long>(1, chunks / 24));
  ks = std::min<long long>(ks, 256);
  const long long per = ((chunks + ks - 1) / ks) * kBK;
  *pslab = per;
  *ksplit = (int)((P + per - 1) / per);
}

}  // namespace

extern "C" int64_t sntc_conv_wgrad_workspace_bytes(int kind, int kh, int kw, int stride, int cin, int cout, int n, int h, int w) {
  WGGeo g;
  if (wg_geometry(kind, kh, kw, stride, cin, cout, h, w, &g) != SNTC_OK || n < 1) return -1;
  int tn, ntm, ntn, ks;
  long long pslab;
  wg_split(g, kh, kw, n, &tn, &ntm, &ntn, &ks, &pslab);
  return ks > 1 ? (int64_t)4 * ks * kh * kw * g.Cs * g.Cd : 0;
}

extern "C" int sntc_conv_wgrad(int kind, int kh, int kw, int stride, int cin, int cout, const float* x, const float* g_out,
                               int n, int h, int w, float* dw, int accumulate, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  if (!x || !g_out || !dw || n < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_wgrad: null argument");
  WGGeo g;
  const int rc = wg_geometry(kind, kh, kw, stride, cin, cout, h, w, &g);
  if (rc != SNTC_OK) return rc;
  WGArgs a{};
  a.S = g.x_is_S ? x : g_out;
  a.D = g.x_is_S ? g_out : x;
  a.n = n; a.Hs = g.Hs; a.Ws = g.Ws; a.Cs = g.Cs; a.Hd = g.Hd; a.Wd = g.Wd; a.Cd = g.Cd;
  a.kh = kh; a.kw = kw; a.stride = stride; a.pt = g.pt; a.pl = g.pl;
  a.M = kh * kw * g.Cs;
  a.N = g.Cd;
  a.P = (long long)n * g.Hd * g.Wd;
  int tn;
  wg_split(g, kh, kw, n, &tn, &a.ntm, &a.ntn, &a.ksplit, &a.pslab);
  const int64_t need = a.ksplit > 1 ? (int64_t)4 * a.ksplit * a.M * a.N : 0;
  if (need > 0 && (!workspace || workspace_bytes < need))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_wgrad: workspace smaller than sntc_conv_wgrad_workspace_bytes()");
  if ((size_t)n * g.Hs * g.Ws * g.Cs >= (1ull << 31) || (size_t)a.P * g.Cd >= (1ull << 31))
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_wgrad: tensors of 2^31 elements or more: split the batch");
  a.slab = a.ksplit > 1 ? static_cast<float*>(workspace) : dw;      // one slab: the kernel writes dW itself
  a.accumulate = accumulate;
#ifdef SNTC_DIAG
  if (const char* e = getenv("SNTC_WG_DBG")) a.dbg = atoi(e);
#endif
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((long long)a.ntm * a.ntn * a.ksplit));
  const bool vec = g.Cs % 4 == 0;
#define SNTC_WG_LAUNCH(WN, TN)                                                                        \
  do {                                                                                                \
    if (vec) hipLaunchKernelGGL((wgrad_kernel<WN, TN, true>), grid, dim3(256), 0, s, a);               \
    else hipLaunchKernelGGL((wgrad_kernel<WN, TN, false>), grid, dim3(256), 0, s, a);                  \
  } while (0)
  switch (tn) {                       // tn holds the column-tile width chosen by wg_split
    case 32: SNTC_WG_LAUNCH(1, 1); break;
    case 64: SNTC_WG_LAUNCH(2, 1); break;
    case 96: SNTC_WG_LAUNCH(1, 3); break;
    case 128: SNTC_WG_LAUNCH(2, 2); break;
    case 160: SNTC_WG_LAUNCH(1, 5); break;
    default: SNTC_WG_LAUNCH(2, 3); break;
  }
#undef SNTC_WG_LAUNCH
  SNTC_HIP(hipGetLastError());
  if (a.ksplit > 1) {
    const long long total = (long long)a.M * a.N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a.slab, total,
                       a.ksplit, dw, accumulate);
    SNTC_HIP(hipGetLastError());
  }
  return SNTC_OK;
}

static int64_t bias_slabs(int64_t npix, int c) {
  const int64_t strips = (c + 63) / 64;
  return std::max<int64_t>(1, std::min<int64_t>((2048 + strips - 1) / strips, (npix + 255) / 256));
}

extern "C" int64_t sntc_bias_grad_workspace_bytes(int64_t npix, int c) {
  if (npix < 1 || c < 1) return -1;
  return 4 * bias_slabs(npix, c) * c;
}

extern "C" int sntc_bias_grad(const float* g, int64_t npix, int c, float* db, int accumulate, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  if (!g || !db || npix < 1 || c < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_bias_grad: bad argument");
  const int64_t slabs = bias_slabs(npix, c);
  if (!workspace || workspace_bytes < 4 * slabs * c) return fail(SNTC_ERR_BAD_SHAPE, "sntc_bias_grad: workspace smaller than sntc_bias_grad_workspace_bytes()");
  const long long pslab = (npix + slabs - 1) / slabs;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((c + 63) / 64, (unsigned)slabs);
  if (c % 4 == 0) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, s, g, (long long)npix, c, pslab, static_cast<float*>(workspace));
  else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, s, g, (long long)npix, c, pslab, static_cast<float*>(workspace));
  SNTC_HIP(hipGetLastError());
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((c + 63) / 64), dim3(256), 0, s, static_cast<const float*>(workspace), (long long)c,
                     (int)slabs, db, accumulate);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

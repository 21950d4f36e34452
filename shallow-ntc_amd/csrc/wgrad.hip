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
// Block = 4 waves (2 x 2), wave tile 64 x (32 TN), MFMA v_mfma_f32_32x32x2_f32 (exact f32).  The operands arrive
// pixel-major ([pixel][channel], NHWC rows), which is exactly [k][m] / [k][n]: the LDS tiles keep that layout and
// every MFMA operand is one ds_read_b32 per lane (lanes 0-31: 32 consecutive channels of pixel k, lanes 32-63 of
// pixel k + 1) -- conflict-free without padding.  Global loads are 16 B per lane along channels, zero outside the
// image (padding taps) or beyond M / N / the slab.
#include <algorithm>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WGArgs {
  const float* S;      // [n, Hs, Ws, Cs]
  const float* D;      // [n, Hd, Wd, Cd]
  float* slab;         // [ksplit][M][N] partial sums
  int n, Hs, Ws, Cs, Hd, Wd, Cd;
  int kh, kw, stride, pt, pl;
  int M, N;            // taps*Cs, Cd
  long long P;         // n*Hd*Wd
  long long pslab;     // pixels per K slab (multiple of BK)
  int ntm, ntn, ksplit;
};

constexpr int kBK = 16;     // pixels per LDS stage
constexpr int kBM = 128;

template <int TN, bool VECS>
__global__ void __launch_bounds__(256) wgrad_kernel(WGArgs a) {
  constexpr int BN = 64 * TN;
  __shared__ float As[2][kBK][kBM];
  __shared__ float Bs[2][kBK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int tm = bid % a.ntm;
  bid /= a.ntm;
  const int tn = bid % a.ntn;
  const int ks = bid / a.ntn;
  const int m0 = tm * kBM, n0 = tn * BN;
  const long long p0 = (long long)ks * a.pslab, p1 = std::min(a.P, p0 + a.pslab);

  // ---- loader roles: A tile = 16 rows x 32 float4 -> 2 per thread; B tile = 16 rows x (BN/4) float4
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
  constexpr int BROWS = 256 / BC4;                          // rows covered per pass (16 for TN=1, 8 for TN=2)
  constexpr int BPASS = kBK / BROWS;
  const int bc4 = tid % BC4, br0 = tid / BC4;
  const int bn = n0 + 4 * bc4;
  const bool bok = bn < a.N;                                // N % 4 == 0 (checked on the host)

  f32x4 ra[2], rb[BPASS];
  auto gload = [&](long long pb) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const long long p = pb + ar0 + 8 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < p1) {
        const long long hw = (long long)a.Hd * a.Wd;
        const int img = (int)(p / hw);
        const int rem = (int)(p - img * hw);
        const int i = rem / a.Wd, j = rem - i * a.Wd;
        if (VECS) {
          const int sy = i * a.stride + aky[0], sx = j * a.stride + akx[0];
          if (aok[0] && sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws)
            v = *reinterpret_cast<const f32x4*>(a.S + (((size_t)img * a.Hs + sy) * a.Ws + sx) * a.Cs + acs[0]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int sy = i * a.stride + aky[e], sx = j * a.stride + akx[e];
            if (aok[e] && sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws)
              v[e] = a.S[(((size_t)img * a.Hs + sy) * a.Ws + sx) * a.Cs + acs[e]];
          }
        }
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const long long p = pb + br0 + BROWS * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < p1 && bok) v = *reinterpret_cast<const f32x4*>(a.D + (size_t)p * a.Cd + bn);
      rb[q] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(&As[buf][ar0 + 8 * q][4 * ac4]) = ra[q];
#pragma unroll
    for (int q = 0; q < BPASS; ++q) *reinterpret_cast<f32x4*>(&Bs[buf][br0 + BROWS * q][4 * bc4]) = rb[q];
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kl = lane >> 5, cl = lane & 31;
  const int arow = wm * 64 + cl, bcol = wn * 32 * TN + cl;
  int buf = 0;
  if (p0 < p1) {
    gload(p0);
    lstore(0);
  }
  __syncthreads();
  for (long long pb = p0; pb < p1; pb += kBK) {
    const bool more = pb + kBK < p1;
    if (more) gload(pb + kBK);
#pragma unroll
    for (int kk = 0; kk < kBK / 2; ++kk) {
      float fa[2], fb[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = As[buf][2 * kk + kl][arow + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = Bs[buf][2 * kk + kl][bcol + 32 * j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  float* out = a.slab + (size_t)ks * a.M * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + cl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + 32 * i + (r >> 2) * 8 + kl * 4 + (r & 3);
        if (row < a.M && col < a.N) out[(size_t)row * a.N + col] = acc[i][j][r];
      }
    }
}

// dW[e] (= or +=) sum over slabs, fixed order
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, long long total, int ksplit,
                                                           float* __restrict__ dw, int accumulate) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    float s = accumulate ? dw[e] : 0.f;
    for (int k = 0; k < ksplit; ++k) s += slab[(size_t)k * total + e];
    dw[e] = s;
  }
}

// db[c] (= or +=) sum over pixels of g[p][c]: one block per 64-channel strip x pixel slab, then ordered reduce
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ g, long long P, int C, long long pslab,
                                                     float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const long long p0 = blockIdx.y * pslab, p1 = std::min(P, p0 + pslab);
  float s = 0.f;
  if (c < C)
    for (long long p = p0 + r; p < p1; p += 4) s += g[(size_t)p * C + c];
  red[r][threadIdx.x & 63] = s;
  __syncthreads();
  if (r == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace sntc

using namespace sntc;

namespace {

struct WGGeo {
  int Hs, Ws, Cs, Hd, Wd, Cd, pt, pl;
  bool x_is_S;
};

// kind: SNTC_CONV2D / SNTC_CONV2D_TRANSPOSE; (h, w) = spatial size of the layer INPUT x
int wg_geometry(int kind, int kh, int kw, int stride, int cin, int cout, int h, int w, WGGeo* g) {
  if (kh < 1 || kw < 1 || stride < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_wgrad: bad sizes");
  if (kind == SNTC_CONV2D) {                       // Keras SAME: out = ceil(in / s), pad_before = pad_total / 2
    const int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
    *g = WGGeo{h, w, cin, ho, wo, cout, std::max((ho - 1) * stride + kh - h, 0) / 2, std::max((wo - 1) * stride + kw - w, 0) / 2, true};
  } else if (kind == SNTC_CONV2D_TRANSPOSE) {      // Keras SAME: out = in * s, pad_before = max(k - s, 0) / 2
    *g = WGGeo{h * stride, w * stride, cout, h, w, cin, std::max(kh - stride, 0) / 2, std::max(kw - stride, 0) / 2, false};
  } else {
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_wgrad: only Conv2D / Conv2DTranspose (Keras SAME) layers are trainable here");
  }
  if (g->Cd % 4) return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_wgrad: the low-resolution side needs a channel count divisible by 4");
  return SNTC_OK;
}

void wg_split(const WGGeo& g, int kh, int kw, int n, int* tn_out, int* ntm, int* ntn, int* ksplit, long long* pslab) {
  const int M = kh * kw * g.Cs, N = g.Cd;
  const int TN = N <= 64 ? 1 : 2;
  const int bn = 64 * TN;
  *tn_out = TN;
  *ntm = (M + kBM - 1) / kBM;
  *ntn = (N + bn - 1) / bn;
  const long long P = (long long)n * g.Hd * g.Wd;
  const long long tiles = (long long)*ntm * *ntn;
  const long long chunks = (P + kBK - 1) / kBK;
  long long ks = std::max<long long>(1, (1536 + tiles - 1) / tiles);       // ~6 blocks per CU in flight
  ks = std::min(ks, std::max<long long>(1, chunks / 4));                    // at least 4 stages per block
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
  return (int64_t)4 * ks * kh * kw * g.Cs * g.Cd;
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
  const int64_t need = (int64_t)4 * a.ksplit * a.M * a.N;
  if (!workspace || workspace_bytes < need) return fail(SNTC_ERR_BAD_SHAPE, "sntc_conv_wgrad: workspace smaller than sntc_conv_wgrad_workspace_bytes()");
  if ((size_t)n * g.Hs * g.Ws * g.Cs >= (1ull << 31) || (size_t)a.P * g.Cd >= (1ull << 31))
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_conv_wgrad: tensors of 2^31 elements or more: split the batch");
  a.slab = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((long long)a.ntm * a.ntn * a.ksplit));
  const bool vec = g.Cs % 4 == 0;
  if (tn == 1) {
    if (vec) hipLaunchKernelGGL((wgrad_kernel<1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_kernel<1, false>), grid, dim3(256), 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((wgrad_kernel<2, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_kernel<2, false>), grid, dim3(256), 0, s, a);
  }
  SNTC_HIP(hipGetLastError());
  const long long total = (long long)a.M * a.N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, s, a.slab, total,
                     a.ksplit, dw, accumulate);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int64_t sntc_bias_grad_workspace_bytes(int64_t npix, int c) {
  if (npix < 1 || c < 1) return -1;
  const int64_t slabs = std::min<int64_t>(256, (npix + 1023) / 1024);
  return 4 * slabs * c;
}

extern "C" int sntc_bias_grad(const float* g, int64_t npix, int c, float* db, int accumulate, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  if (!g || !db || npix < 1 || c < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_bias_grad: bad argument");
  const int64_t slabs = std::min<int64_t>(256, (npix + 1023) / 1024);
  if (!workspace || workspace_bytes < 4 * slabs * c) return fail(SNTC_ERR_BAD_SHAPE, "sntc_bias_grad: workspace smaller than sntc_bias_grad_workspace_bytes()");
  const long long pslab = (npix + slabs - 1) / slabs;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_kernel, dim3((c + 63) / 64, (unsigned)slabs), dim3(256), 0, s, g, (long long)npix, c, pslab,
                     static_cast<float*>(workspace));
  SNTC_HIP(hipGetLastError());
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((c + 255) / 256), dim3(256), 0, s, static_cast<const float*>(workspace), (long long)c,
                     (int)slabs, db, accumulate);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

// up_small.hip -- the LAST layer of the multi-layer syntheses: a 5 x 5 / 2 (or 9 x 9 / 4) transposed convolution down to the 3 image channels.
//   reference common/transforms.py:172-175  MBT2018Synthesis: tfc.SignalConv2D(3, (5, 5), corr=False, strides_up=2, "same_zeros")
//             common/transforms.py:195-206  CNNSynthesis:     conv_t_k5s2(output_channels) = Keras Conv2DTranspose(3, 5, strides=2, "SAME") (:85-87)
//             common/transforms.py:131-134  BLS2017Synthesis: tfc.SignalConv2D(3, (9, 9), corr=False, strides_up=4, "same_zeros")
//             (BASELINE configs[1], mshyper/configs/mbt2018.py, is the first of the three; configs[0], factorized/configs/bls2017.py, the last)
//
// With three output channels the layer is no GEMM worth the name: on the gather GEMM its four output phases are four groups of
// N = 3 columns in 32-wide MFMA tiles (10 % of the MFMAs do work): 0.38 ms = 9.9 TFLOP/s for 8 x 128 x 128 x 192 -> 8 x 256 x 256 x 3,
// a quarter of that config's decode.  It is an HBM read stream (768 B in per 48 B out) with 6.25 x cin x 3 MACs per output pixel,
// which the vector ALU does faster than a 90 %-empty matrix tile -- the structure of two_layer_tail_kernel's output stage (pixel.hip),
// with the input channels walked in 16-channel slabs so that any cin % 16 == 0 fits:
//   * a block owns 16 x 16 macro pixels q and their 2 x 2 output quads o = 2 q + phi - pt (phi in {0, 1}^2; pt = 1: Keras SAME,
//     pt = 2: SignalConv2D's centred kernel -- SURVEY.md A.2 / A.3), one thread per OUTPUT pixel, one phase phi per wave; phase 0
//     of an axis uses kernel indices {0, 2, 4} from source rows q, q - 1, q - 2, phase 1 uses {1, 3} from q, q - 1 (gather form of
//     the scatter out[2 i + k - pt] += x[i] w[k]);
//   * per slab the 18 x 18 x 16 input tile sits in LDS (zeros outside the image); a thread reads its phase's 9 / 6 / 6 / 4 source
//     pixels four channels at a time and runs taps x 3 outputs on them; the next slab's tile travels global -> registers under the
//     arithmetic;
//   * the packed weights [slab][channel quad][tap][out][4] arrive by scalar loads (uniform addresses): SGPR operands, no LDS, no vector loads.
// Arithmetic: fp32 fma chains per 16-channel slab (channel quads, taps row-major inside), the slabs' sums added in order -- NOT the
// gather GEMM's order; the layer is
// decoder-only and its chains do not depend on the batch (an image alone == the image in a batch); tested against the float64 oracle.
#include <algorithm>
#include <type_traits>
#include "sntc_internal.h"
#include "device_math.h"

namespace sntc {
namespace {

constexpr int kTQ = 16, kTH = kTQ + 2;        // macro pixels per block side, input tile side (two halo rows / columns in front)
constexpr int kCS = 16;                       // channels per slab
constexpr int kPX = 20;                       // LDS words per tile pixel: 16 + 4, an odd number of 16-B slots (conflict-free row reads)
constexpr int kROW = 384;                     // LDS words per tile row (>= 18 * 20, a multiple of 64)
constexpr int kTileLoads = (kTH * kTH * (kCS / 4) + 1023) / 1024;    // 16-B loads per thread and slab (1024 threads)

struct UpArgs {
  const float* x;          // [n, h, w, cin]
  float* y;                // [n, 2h, 2w, CO]
  const float* wpack;      // [cin / 16][4 channel quads][25 taps][CO][4]
  const float* bias;       // [CO] (zeros where the layer has none)
  int h, w, cin, pt;
};

constexpr int taps_of(int ks, int s, int phase) { return (ks - phase + s - 1) / s; }     // kernel indices phase, phase + s, ... < ks

// One output PHASE per wave: a block is 16 x 16 macro pixels x 4 phases = 1024 threads, wave w computes phase (w / 8, (w / 4) & 1) of
// macro rows 4 (w & 3) .. + 3.  A wave then needs only its phase's taps (9 / 6 / 6 / 4 of the 25), i.e. a quarter of the scalar weight
// loads per wave at four times the waves.  (Measured, 8 x 128 x 128 x 192: one thread = one macro pixel = all four phases, 256
// threads: 0.151 ms -- it waited for 1200 scalar loads per slab and wave with 2.5 waves per SIMD to hide them behind; this form:
// 0.116 ms; two phases per wave paired to 13 / 12 taps, 512 threads, four blocks per CU: 0.127 ms.  The scalar path stays the limit.)
// KS x KS / S: 5 x 5 / 2 (a wave = one of the 4 phases) or 9 x 9 / 4 (a wave = one phase ROW py, its four px phases per thread: 27 or
// 18 of the 81 taps) -- either way 4 kinds of waves x 4 groups of macro rows.
template <int KS, int S, int CO>
__global__ void __launch_bounds__(1024) up_small_kernel(const UpArgs a) {
  constexpr int NPX = S == 2 ? 1 : S;                // px phases per thread
  static_assert((S == 2 && KS == 5) || (S == 4 && KS == 9), "two source pixels of halo, four kinds of waves");
  __shared__ __attribute__((aligned(16))) float sh[kTH * kROW];
  const int img = blockIdx.z;
  const int qy0 = blockIdx.y * kTQ, qx0 = blockIdx.x * kTQ;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kind = wave >> 2;                        // uniform per wave: S = 2: (py, px) = (kind >> 1, kind & 1); S = 4: py = kind
  const int ty = 4 * (wave & 3) + ((tid & 63) >> 4), tx = tid & 15;
  const int nslab = a.cin / kCS;

  // this thread's share of a slab's tile: 16-B chunk `c4` of tile pixel `p` (zeros outside the image); 1296 chunks, 1024 threads
  int lds_off[kTileLoads];
  int64_t src_off[kTileLoads];                // float offset of (pixel, chunk) in x for slab 0, or -1
#pragma unroll
  for (int i = 0; i < kTileLoads; ++i) {
    const int idx = tid + 1024 * i;
    const int p = idx >> 2, c4 = idx & 3;
    const int ly = p / kTH, lx = p - ly * kTH;
    const int iy = qy0 - 2 + ly, ix = qx0 - 2 + lx;
    const bool in_tile = p < kTH * kTH;
    const bool ok = in_tile && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
    lds_off[i] = in_tile ? ly * kROW + lx * kPX + 4 * c4 : -1;
    src_off[i] = ok ? (((int64_t)img * a.h + iy) * a.w + ix) * a.cin + 4 * c4 : -1;
  }
  f32x4 R[kTileLoads];
  auto load_slab = [&](int cc) {
#pragma unroll
    for (int i = 0; i < kTileLoads; ++i)
      R[i] = src_off[i] >= 0 ? *reinterpret_cast<const f32x4*>(a.x + src_off[i] + cc * kCS) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int i = 0; i < kTileLoads; ++i)
      if (lds_off[i] >= 0) *reinterpret_cast<f32x4*>(sh + lds_off[i]) = R[i];
  };

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  float acc[NPX][CO];
#pragma unroll
  for (int i = 0; i < NPX; ++i)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[i][o] = a.bias[o];

  // one slab of this wave's phases (PY, PX0 .. PX0 + NPX - 1): taps ky = PY + S jy, kx = px + S jx from tile pixel
  // (ty + 2 - jy, tx + 2 - jx).  Two-level summation: the slab's products in two interleaved chains per output (even / odd channel of
  // a pair, v_pk_fma_f32: the pixel's channel pair is a VGPR pair as it comes from LDS, the weight pair an SGPR pair as it comes from
  // the scalar load), the slabs' sums added in slab order (a single chain over 2000 products at cin = 320 carried 4.7 x the MFMA
  // path's rounding error)
  auto slab = [&](auto PYc, auto PX0c, const float* wslab) {
    constexpr int PY = decltype(PYc)::value, PX0 = decltype(PX0c)::value;
    f32x2 part[NPX][CO];
#pragma unroll
    for (int i = 0; i < NPX; ++i)
#pragma unroll
      for (int o = 0; o < CO; ++o) part[i][o] = f32x2{0.0f, 0.0f};
#pragma unroll 1
    for (int c4 = 0; c4 < kCS / 4; ++c4) {
      const float* wq = wslab + c4 * (KS * KS * CO * 4);   // [tap][out][4 channels of this quad]: 12 consecutive floats per tap
#pragma unroll
      for (int jy = 0; jy < taps_of(KS, S, PY); ++jy)
#pragma unroll
        for (int jx = 0; jx < taps_of(KS, S, PX0); ++jx) {         // (the first px phase has the most taps)
          const f32x4 hv = *reinterpret_cast<const f32x4*>(sh + (ty + 2 - jy) * kROW + (tx + 2 - jx) * kPX + 4 * c4);
          const f32x2 h01 = f32x2{hv[0], hv[1]}, h23 = f32x2{hv[2], hv[3]};
#pragma unroll
          for (int i = 0; i < NPX; ++i) {
            if (jx < taps_of(KS, S, PX0 + i)) {
              const int ky = PY + S * jy, kx = PX0 + i + S * jx;
              const float* wp = wq + (ky * KS + kx) * (CO * 4);
#pragma unroll
              for (int o = 0; o < CO; ++o) {
                const f32x2 w01 = f32x2{wp[o * 4], wp[o * 4 + 1]}, w23 = f32x2{wp[o * 4 + 2], wp[o * 4 + 3]};
                part[i][o] = __builtin_elementwise_fma(h01, w01, part[i][o]);
                part[i][o] = __builtin_elementwise_fma(h23, w23, part[i][o]);
              }
            }
          }
        }
    }
#pragma unroll
    for (int i = 0; i < NPX; ++i)
#pragma unroll
      for (int o = 0; o < CO; ++o) acc[i][o] += part[i][o][0] + part[i][o][1];
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  load_slab(0);
  for (int cc = 0; cc < nslab; ++cc) {
    __syncthreads();                                   // everybody has left the previous slab's tile
    store_slab();
    __syncthreads();
    if (cc + 1 < nslab) load_slab(cc + 1);             // travels under this slab's arithmetic
    const float* wslab = a.wpack + (size_t)cc * (KS * KS * CO * kCS);      // uniform: the weights are scalar loads
    if constexpr (S == 2) {
      if (kind == 0) slab(I0{}, I0{}, wslab);
      else if (kind == 1) slab(I0{}, I1{}, wslab);
      else if (kind == 2) slab(I1{}, I0{}, wslab);
      else slab(I1{}, I1{}, wslab);
    } else {
      if (kind == 0) slab(I0{}, I0{}, wslab);
      else if (kind == 1) slab(I1{}, I0{}, wslab);
      else if (kind == 2) slab(I2{}, I0{}, wslab);
      else slab(I3{}, I0{}, wslab);
    }
  }

  const int py = S == 2 ? kind >> 1 : kind, px0 = S == 2 ? kind & 1 : 0;
  const int oy = S * (qy0 + ty) + py - a.pt;
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int ox = S * (qx0 + tx) + px0 + i - a.pt;
    if ((unsigned)oy < (unsigned)(S * a.h) && (unsigned)ox < (unsigned)(S * a.w)) {
      float* dst = a.y + (((int64_t)img * (S * a.h) + oy) * (S * a.w) + ox) * CO;
#pragma unroll
      for (int o = 0; o < CO; ++o) dst[o] = acc[i][o];
    }
  }
}

// wpack[slab][channel quad][tap][o][4] from the layer's kernel: Keras Conv2DTranspose [5, 5, Cout, Cin] (io == 0) or tfc.SignalConv2D
// [5, 5, Cin, Cout] (io == 1: true convolution, no flip in scatter form -- SURVEY.md A.3)
__global__ void __launch_bounds__(256) up_small_pack_kernel(const float* __restrict__ w, float* __restrict__ wpack, int cin, int co, int io, int ntap) {
  const int total = (cin / kCS) * ntap * co * kCS;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 3;
    int r = idx >> 2;
    const int o = r % co; r /= co;
    const int tap = r % ntap; r /= ntap;
    const int c4 = r % (kCS / 4);
    const int slab = r / (kCS / 4);
    const int c = slab * kCS + 4 * c4 + e;
    wpack[idx] = io ? w[((size_t)tap * cin + c) * co + o] : w[((size_t)tap * co + o) * cin + c];
  }
}

__global__ void up_small_bias_kernel(const float* b, float* out, int co) {
  const int i = threadIdx.x;
  if (i < co) out[i] = b ? b[i] : 0.0f;
}

}  // namespace
}  // namespace sntc

struct sntc_upsmall_plan {
  int kind = SNTC_CONV2D_TRANSPOSE, cin = 0, cout = 0, k = 5, stride = 2;
  float* wpack = nullptr;
  float* bias = nullptr;
};

using namespace sntc;

static int up_pack(sntc_upsmall_plan* p, const float* w, const float* bias, hipStream_t s) {
  const int total = (p->cin / kCS) * p->k * p->k * p->cout * kCS;
  hipLaunchKernelGGL(up_small_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, p->wpack, p->cin, p->cout,
                     p->kind == SNTC_SIGNAL_UP ? 1 : 0, p->k * p->k);
  hipLaunchKernelGGL(up_small_bias_kernel, dim3(1), dim3(64), 0, s, bias, p->bias, p->cout);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "small-output transposed convolution: weight packing");
  return SNTC_OK;
}

extern "C" int sntc_upsmall_supported(int kind, int k, int stride, int cin, int cout) {
  return ((kind == SNTC_CONV2D_TRANSPOSE || kind == SNTC_SIGNAL_UP) && ((k == 5 && stride == 2) || (k == 9 && stride == 4)) && cin >= kCS && cin % kCS == 0 && cout == 3) ? 1 : 0;
}

static void up_free(sntc_upsmall_plan* p) {
  if (p->wpack) (void)hipFree(p->wpack);
  if (p->bias) (void)hipFree(p->bias);
  delete p;
}

extern "C" int sntc_upsmall_plan_create(int kind, int k, int stride, int cin, int cout, const float* w, const float* bias, void* stream,
                                        sntc_upsmall_plan** plan) {
  if (!plan || !w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_upsmall_plan_create: null argument");
  if (!sntc_upsmall_supported(kind, k, stride, cin, cout))
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_upsmall_plan_create: the kernel exists for Conv2DTranspose / SignalConv2D(strides_up) 5 x 5 / 2 and 9 x 9 / 4, "
                                      "cin % 16 == 0, 3 output channels");
  auto* p = new sntc_upsmall_plan();
  p->kind = kind; p->cin = cin; p->cout = cout; p->k = k; p->stride = stride;
  if (hipMalloc(&p->wpack, sizeof(float) * k * k * cin * cout) != hipSuccess || hipMalloc(&p->bias, sizeof(float) * 4) != hipSuccess) {
    up_free(p);
    return fail(SNTC_ERR_HIP, "sntc_upsmall_plan_create: out of device memory");
  }
  if (int rc = up_pack(p, w, bias, (hipStream_t)stream)) {
    up_free(p);
    return rc;
  }
  *plan = p;
  return SNTC_OK;
}

extern "C" int sntc_upsmall_plan_update(sntc_upsmall_plan* p, const float* w, const float* bias, void* stream) {
  if (!p || !w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_upsmall_plan_update: null argument");
  return up_pack(p, w, bias, (hipStream_t)stream);
}

extern "C" void sntc_upsmall_plan_destroy(sntc_upsmall_plan* p) {
  if (p) up_free(p);
}

extern "C" int64_t sntc_upsmall_flops(const sntc_upsmall_plan* p, int n, int h, int w) {
  if (!p || n < 0 || h < 0 || w < 0) return -1;
  return 2 * (int64_t)n * h * w * p->k * p->k * p->cin * p->cout;
}

extern "C" int sntc_upsmall_forward(const sntc_upsmall_plan* p, const float* x, int n, int h, int w, float* y, void* stream) {
  if (!p) return fail(SNTC_ERR_BAD_SHAPE, "sntc_upsmall_forward: null plan");
  if (n < 0 || h < 0 || w < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_upsmall_forward: negative size");
  if (n == 0 || h == 0 || w == 0) return SNTC_OK;
  if (!x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_upsmall_forward: null argument");
  if (n > 65535) return fail(SNTC_ERR_BAD_SHAPE, "sntc_upsmall_forward: more than 65535 images per call; split the batch");
  UpArgs a{};
  a.x = x; a.y = y; a.wpack = p->wpack; a.bias = p->bias;
  a.h = h; a.w = w; a.cin = p->cin;
  a.pt = p->kind == SNTC_SIGNAL_UP ? (p->k - 1) / 2 : (p->k - p->stride) / 2;     // the centred kernel; A.2: Keras SAME
  const dim3 grid((w + 1 + kTQ - 1) / kTQ, (h + 1 + kTQ - 1) / kTQ, n);
  if (p->stride == 2) hipLaunchKernelGGL((up_small_kernel<5, 2, 3>), grid, dim3(1024), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((up_small_kernel<9, 4, 3>), grid, dim3(1024), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "small-output transposed convolution launch");
  return SNTC_OK;
}

// rgb_conv.hip -- the RGB first layer of the analysis transforms (reference common/elic.py:147 / :253-270 build_conv,
// common/transforms.py:183: Keras Conv2D(k = 5, strides = 2, padding = "SAME") on 3 input channels) as a kernel of its own.
//
// Through round 5 this layer ran on the gather GEMM as a "row-packed" plan (one kernel ROW = one 16-deep K stage) over a
// zero-padded copy of the image: 0.69 + 0.05 ms for 18 x 512 x 768, i.e. 74 TFLOP/s of arithmetic and 2 TB/s of output -- under
// neither roof (K = 80: a tile is five stages long and its bookkeeping costs as much as its MFMAs).  The layer is
// really a STORE stream (4.6 KB written per 12 B read) with 75 MACs per output value, so here:
//   * a workgroup (512 threads = 8 waves, one per CU) is persistent; each WAVE owns 32 consecutive output pixels of one output row
//     at a time and ALL output channels (cout / 32 accumulator tiles of 32 x 32; pixels = MFMA A operand, weights = B operand: a lane
//     ends up with ONE channel of 16 pixels, so a dword store per register writes whole 128-B lines without any shuffle);
//   * the whole packed weight matrix (k stages x cout rows x 16 = 61 KB for 192 channels) sits in LDS for the life of the workgroup
//     -- the only thing its waves share: no barrier after it has landed;
//   * a unit's input patch (k rows of ((32 - 1) s + k) cin floats: 4 KB) is wave-private and double-buffered in LDS -- the next
//     unit's patch travels (global -> registers -> LDS) under this unit's MFMAs; SAME padding = zeros written into the patch, no
//     padded copy of the image;
//   * a pixel fragment is FOUR CONSECUTIVE floats of a patch row: slot ci = kx cin + c of kernel row ky for output pixel p lives
//     at float (s p + kx) cin + c = s cin p + ci of patch row s r + ky -- the row-packed K order needs no gather at all;
//   * the result leaves with bias + activation, 96 line-covering dword stores per unit and wave.
// Every output element is the SAME k-ordered fp32 fma chain as the row-packed plan's (stage = kernel row; MFMA e of k-group g sums
// slots {8 g + e, 8 g + 4 + e}; slot 15 and out-of-image pixels contribute fma(x, 0, acc) / fma(0, w, acc)): bit-identical to it.
#include <algorithm>
#include <cmath>
#include <mutex>
#include "sntc_internal.h"

#include "rb_common.h"

namespace sntc {
using namespace rb;

namespace {

#define RGB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int kTW = 32;                   // a unit: 32 consecutive output pixels of one output row (= one MFMA fragment), one wave
constexpr int kMaxK = 5;                  // kernel rows (K stages) held in LDS
constexpr int kCin = 3, kStride = 2;      // the reference's first layers: RGB, stride 2
constexpr int kLS = kStride * kCin;       // floats between neighbouring output pixels in a patch row
constexpr int kPRowF = 204;               // floats per patch row: (31 s + k) cin = 201 used, fragment reads reach 6 * 31 + 15 = 201
constexpr int kPatchF = kMaxK * kPRowF;   // a unit's patch: k image rows, 1020 floats
constexpr int kPLoads = (kPatchF + 63) / 64;                        // dwords per lane and unit
constexpr int kWaves = 8;

struct RGBArgs {
  const float* x;          // [n, H, W, 3]
  float* y;                // [n, Ho, Wo, cout]
  const float* wpack;      // [k][cout][16], LDS image order (16-B chunks XOR-swizzled by (row >> 2) & 3)
  const float* bias;       // [cout] (zeros where the layer has none)
  unsigned xbytes, ybytes;
  int N, H, W, Ho, Wo;
  int k, pt, pl;
  int tiles_x, nunits;
  float act_m, act_c;
};

template <int NT>
struct RGBCfg {
  static constexpr int COUT = NT * 32;
  static constexpr int UNIT = COUT * 16;                                    // floats per K stage
  static constexpr size_t LDS = (size_t)(kMaxK * UNIT + kWaves * 2 * kPatchF + COUT) * 4;
};

// The waves of a workgroup share nothing but the weights: every wave stages its OWN patch (k image rows, double-buffered, 8 KB)
// and walks its own contiguous range of units, so there is no workgroup barrier after the weights have landed and the two waves
// of a SIMD drift apart -- one stores its results while the other multiplies.  (Measured on the way, tools/rgb_conv_block.py: a
// tile per workgroup with a barrier per tile, this structure with a 4 x 4 quad transpose in front of 16-B stores, and this one with
// dword stores all took 0.53 - 0.56 ms for 18 x 512 x 768 -- neither the lockstep nor the store pattern decided the time.  The VECTOR
// instruction count next to the MFMAs did -- the fp32 MFMA and the vector ALU share their multipliers: per unit, 16 patch loads and 96
// stores took ~300 address / select instructions, now ~20 (scalar bases, per-lane constant offsets, uniform fast paths for interior
// units): 0.48 ms; the row-packed plan + its padding pass: 0.80 ms.  DESIGN.md 4.1f.)
// KH: kernel rows as a compile-time constant (5: the ten K steps of a unit are one scheduled block, the fragments of step s + 1 read
// under the MFMAs of step s) or 0 (any k <= 5 at run time: a plain loop).
// ACT: an activation follows the bias (false: none -- the ELIC first layer -- and the epilogue is one addition per value).
template <int NT, int KH, bool ACT>
__global__ void __launch_bounds__(512, 2) rgb_conv_kernel(const RGBArgs a) {
  using K = RGBCfg<NT>;
  constexpr int COUT = K::COUT, UNIT = K::UNIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = reinterpret_cast<float*>(smem);            // [k][COUT][16]
  float* patches = wl + kMaxK * UNIT;                    // [wave][2][kMaxK][kPRowF]
  float* lbias = patches + kWaves * 2 * kPatchF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.ybytes, 0x00020000);

  // ---- once per workgroup: the packed weights and the bias into LDS
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack);
    f32x4* dst = reinterpret_cast<f32x4*>(wl);
    const int nv = a.k * UNIT / 4;
    for (int i = tid; i < nv; i += 512) dst[i] = src[i];
    for (int i = tid; i < COUT; i += 512) lbias[i] = a.bias[i];
  }
  __syncthreads();

  // this wave's contiguous range of units (wave-level persistence: 8 x gridDim.x workers)
  const int GW = (int)gridDim.x * kWaves, gw = (int)blockIdx.x * kWaves + wave;
  const int u_lo = (int)((long long)a.nunits * gw / GW);
  const int u_hi = (int)((long long)a.nunits * (gw + 1) / GW);
  if (u_lo >= u_hi) return;

  // this lane's share of a patch: linear float index lane + 64 i -> (kernel row, float in row).  lofs: its byte offset from the
  // patch's first float in the image (beyond the buffer for rows >= k: those read zeros whatever the scalar base is -- the
  // range check of a raw buffer covers the vector offset only)
  const int rowf = a.W * kCin;                          // floats per image row
  unsigned lofs[kPLoads];
#pragma unroll
  for (int i = 0; i < kPLoads; ++i) {
    const int idx = lane + 64 * i;
    const int pr = idx / kPRowF, pc = idx - pr * kPRowF;
    lofs[i] = pr < a.k ? (unsigned)(pr * rowf + pc) * 4u : kOOB;
  }
  auto coords = [&](int unit, int* tn, int* oy, int* tx0) {
    const int per = a.tiles_x * a.Ho;
    *tn = unit / per;
    const int r = unit - *tn * per;
    *oy = r / a.tiles_x;
    *tx0 = (r - *oy * a.tiles_x) * kTW;
  };
  float R[kPLoads];
  auto load_patch = [&](int unit) {
    int tn, oy, tx0;
    coords(unit, &tn, &oy, &tx0);
    const int iy0 = oy * kStride - a.pt;
    const int fx0 = (tx0 * kStride - a.pl) * kCin;      // first float of the patch inside its image row (may be negative)
    if (iy0 >= 0 && iy0 + a.k <= a.H && fx0 >= 0 && fx0 + kPRowF <= rowf) {
      // the whole patch lies inside the image (all but the border units): a scalar base + the lane's constant offsets
      const int base = ((tn * a.H + iy0) * rowf + fx0) * 4;
#pragma unroll
      for (int i = 0; i < kPLoads; ++i)
        R[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xs, (int)lofs[i], base, 0));
    } else {
#pragma unroll
      for (int i = 0; i < kPLoads; ++i) {
        const int idx = lane + 64 * i;
        const int pr = idx / kPRowF, pc = idx - pr * kPRowF;
        const int iy = iy0 + pr, fx = fx0 + pc;
        const bool ok = pr < a.k && (unsigned)iy < (unsigned)a.H && (unsigned)fx < (unsigned)rowf;
        const unsigned off = ok ? (unsigned)((tn * a.H + iy) * rowf + fx) * 4u : kOOB;
        R[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xs, (int)off, 0, 0));
      }
    }
  };
  float* mypatch = patches + wave * 2 * kPatchF;
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kPLoads; ++i)
      if (lane + 64 * i < kPatchF) mypatch[buf * kPatchF + lane + 64 * i] = R[i];
  };
  load_patch(u_lo);
  store_patch(0);

  // fragment addressing
  const int swz = (l31 >> 2) & 3;
  const int woff0 = l31 * 16 + (((0 + h) ^ swz) << 2);   // k-group 0: slots 4 h .. 4 h + 3
  const int woff1 = l31 * 16 + (((2 + h) ^ swz) << 2);   // k-group 1: slots 8 + 4 h ..
  const int poff_lane = kLS * l31 + 4 * h;
  unsigned ovo[16];                                      // register r of a tile -> byte offset of (its pixel, this lane's channel) in the unit
#pragma unroll
  for (int r = 0; r < 16; ++r) ovo[r] = (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * h) * COUT + l31) * 4u;
  float bj[NT];                                          // this lane's channel of every tile
#pragma unroll
  for (int j = 0; j < NT; ++j) bj[j] = lbias[32 * j + l31];

  int cur = 0;
  for (int unit = u_lo; unit < u_hi; ++unit) {
    int n, oy, x0;
    coords(unit, &n, &oy, &x0);
    const bool more = unit + 1 < u_hi;
    if (more) load_patch(unit + 1);                      // travels under this unit's MFMAs

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const float* pb = mypatch + cur * kPatchF + poff_lane;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // the fragments of K step st = 2 ky + g: weights of all NT channel tiles (slots 8 g + 4 h ..), four pixels' worth of patch floats
    auto read_frag = [&](f32x4 (&Fw)[NT], f32x4& P, int st) {
      const int ky = st >> 1, g = st & 1;
      const float* wrow = wl + ky * UNIT + (g ? woff1 : woff0);
      const float* prow_ = pb + ky * kPRowF + 8 * g;
      const f32x2 p0 = *reinterpret_cast<const f32x2*>(prow_);
      const f32x2 p1 = *reinterpret_cast<const f32x2*>(prow_ + 2);
      P = f32x4{p0[0], p0[1], p1[0], p1[1]};
#pragma unroll
      for (int j = 0; j < NT; ++j) Fw[j] = *reinterpret_cast<const f32x4*>(wrow + j * 32 * 16);
    };
    if constexpr (KH > 0 && NT <= 6) {
      f32x4 FwA[NT], FwB[NT], PA, PB;
      read_frag(FwA, PA, 0);
      static_for<0, 2 * KH>([&](auto S) {
        constexpr int st = decltype(S)::value;
        f32x4(&Fc)[NT] = (st & 1) ? FwB : FwA;
        f32x4(&Fn)[NT] = (st & 1) ? FwA : FwB;
        f32x4& Pc = (st & 1) ? PB : PA;
        f32x4& Pn = (st & 1) ? PA : PB;
        if constexpr (st + 1 < 2 * KH) read_frag(Fn, Pn, st + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = RGB_MFMA(Pc[e], Fc[j][e], acc[j]);
        // the next step's fragment reads behind the step's first MFMA (mask 0x8 MFMA, 0x100 DS read)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (st + 1 < 2 * KH) __builtin_amdgcn_sched_group_barrier(0x100, NT + 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 1, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (KH > 0) {
      // 256 output channels: 128 accumulator registers leave no room for a second set of fragments (26 spilled registers with it)
      static_for<0, 2 * KH>([&](auto S) {
        f32x4 Fw[NT], P;
        read_frag(Fw, P, decltype(S)::value);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = RGB_MFMA(P[e], Fw[j][e], acc[j]);
      });
    } else {
#pragma unroll 1
      for (int st = 0; st < 2 * a.k; ++st) {
        f32x4 Fw[NT], P;
        read_frag(Fw, P, st);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = RGB_MFMA(P[e], Fw[j][e], acc[j]);
      }
    }
    if (more) store_patch(cur ^ 1);

    // ---- epilogue.  Pixels are the MFMA's A operand, so register r of tile jt, lane l is channel 32 jt + l % 32 of pixel
    // (r & 3) + 8 (r >> 2) + 4 (l / 32): a dword store per register writes two whole 128-B lines (two pixels x 32 consecutive
    // channels) with no shuffle at all -- the lane = pixel layout of csrc/rb_fused.hip needs a 4 x 4 quad transpose (64 vector
    // instructions per channel tile) before its stores cover whole lines, and vector instructions are what this kernel is short
    // of: the fp32 MFMA and the vector ALU share their multipliers (mfma_valu.hip).
    const int obase = ((n * a.Ho + oy) * a.Wo + x0) * (COUT * 4);         // scalar: the unit's first output pixel
    // gather_gemm_kernel.h::apply_act without a branch: max(v, m v + c) with (m, c) = (0, -inf) none, (0, +0) relu = max(v, 0),
    // (0.2, -0) leaky relu = v >= 0 ? v : 0.2 v  (x + -0 = x; 0 v + -inf = -inf for finite v) -- the same values, signed zeros included
    auto emit = [&](const unsigned (&po)[16]) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[jt][r] + bj[jt];
          if constexpr (ACT) v = fmaxf(v, a.act_m * v + a.act_c);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, (int)po[r], obase + jt * 128, 0);
        }
      }
    };
    if (x0 + kTW <= a.Wo) {
      emit(ovo);                                         // all 32 pixels inside the row: the lane's constant offsets as they are
    } else {
      unsigned po[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) po[r] = (x0 + (r & 3) + 8 * (r >> 2) + 4 * h < a.Wo) ? ovo[r] : kOOB;
      emit(po);
    }
    cur ^= 1;
  }
}

// wpack[ky][row][16] in the LDS image order, from the Keras HWIO kernel w[kh, kw, cin, cout]
__global__ void __launch_bounds__(256) rgb_pack_kernel(const float* __restrict__ w, float* __restrict__ wpack, int k, int cin, int cout) {
  const int total = k * cout * 16;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ky = idx / (cout * 16);
    const int rem = idx - ky * cout * 16;
    const int row = rem >> 4, pos = rem & 15;
    const int chunk = (pos >> 2) ^ ((row >> 2) & 3);
    const int ci = chunk * 4 + (pos & 3);                 // slot of the stage: (kx, c) = (ci / cin, ci % cin), zeros behind k cin
    float v = 0.0f;
    if (ci < k * cin) {
      const int kx = ci / cin, c = ci - kx * cin;
      v = w[((size_t)(ky * k + kx) * cin + c) * cout + row];
    }
    wpack[idx] = v;
  }
}

__global__ void rgb_bias_kernel(const float* b, float* out, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) out[i] = b ? b[i] : 0.0f;
}

constexpr int kMaxDev = 16;
struct RGBDevice {
  std::once_flag once;
  int rc = SNTC_OK;
  int num_cus = 0;
};
RGBDevice g_rgbdev[kMaxDev];

int rgb_init(int* num_cus) {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the first-layer tables");
  RGBDevice& D = g_rgbdev[dev];
  std::call_once(D.once, [&] {
    D.rc = [&]() -> int {
      hipDeviceProp_t prop;
      SNTC_HIP(hipGetDeviceProperties(&prop, dev));
      D.num_cus = prop.multiProcessorCount;
#define RGB_ATTR1(NT, KH, ACT) SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rgb_conv_kernel<NT, KH, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RGBCfg<NT>::LDS))
#define RGB_ATTR(NT, KH) RGB_ATTR1(NT, KH, false); RGB_ATTR1(NT, KH, true)
      RGB_ATTR(4, 5); RGB_ATTR(6, 5); RGB_ATTR(8, 5); RGB_ATTR(4, 0); RGB_ATTR(6, 0); RGB_ATTR(8, 0);
#undef RGB_ATTR
#undef RGB_ATTR1
      return SNTC_OK;
    }();
  });
  *num_cus = D.num_cus;
  return D.rc;
}

}  // namespace
}  // namespace sntc

struct sntc_rgbconv_plan {
  int k = 0, cout = 0, act = 0, kind = SNTC_CONV2D;
  float* wpack = nullptr;
  float* bias = nullptr;
  int max_workgroups = 0;
};

using namespace sntc;

static int rgb_pack(sntc_rgbconv_plan* p, const float* w, const float* bias, hipStream_t s) {
  hipLaunchKernelGGL(rgb_pack_kernel, dim3((p->k * p->cout * 16 + 255) / 256), dim3(256), 0, s, w, p->wpack, p->k, kCin, p->cout);
  hipLaunchKernelGGL(rgb_bias_kernel, dim3((p->cout + 255) / 256), dim3(256), 0, s, bias, p->bias, p->cout);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "first-layer weight packing");
  return SNTC_OK;
}

extern "C" int sntc_rgbconv_supported(int kind, int k, int stride, int cin, int cout, int act) {
  return ((kind == SNTC_CONV2D || kind == SNTC_SIGNAL_DOWN) && cin == kCin && stride == kStride && k >= 1 && k <= kMaxK && k * cin <= 16 && (cout == 128 || cout == 192 || cout == 256) &&
          (act == SNTC_ACT_NONE || act == SNTC_ACT_RELU || act == SNTC_ACT_LEAKY_RELU)) ? 1 : 0;
}

static void rgb_free(sntc_rgbconv_plan* p) {
  if (p->wpack) (void)hipFree(p->wpack);
  if (p->bias) (void)hipFree(p->bias);
  delete p;
}

extern "C" int sntc_rgbconv_plan_create(int kind, int k, int stride, int cin, int cout, const float* w, const float* bias, int act,
                                        void* stream, sntc_rgbconv_plan** plan) {
  if (!plan || !w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_plan_create: null argument");
  if (!sntc_rgbconv_supported(kind, k, stride, cin, cout, act))
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_rgbconv_plan_create: the first-layer kernel exists for Conv2D / SignalConv2D(corr, strides_down), k <= 5, stride 2, 3 -> 128 / 192 / 256 channels, "
                                      "no activation / relu / leaky relu");
  int cus = 0;
  if (int rc = rgb_init(&cus)) return rc;
  auto* p = new sntc_rgbconv_plan();
  p->k = k; p->cout = cout; p->act = act; p->kind = kind;
  if (hipMalloc(&p->wpack, sizeof(float) * k * cout * 16) != hipSuccess || hipMalloc(&p->bias, sizeof(float) * cout) != hipSuccess) {
    rgb_free(p);
    return fail(SNTC_ERR_HIP, "sntc_rgbconv_plan_create: out of device memory");
  }
  if (int rc = rgb_pack(p, w, bias, (hipStream_t)stream)) {
    rgb_free(p);
    return rc;
  }
  *plan = p;
  return SNTC_OK;
}

extern "C" int sntc_rgbconv_plan_update(sntc_rgbconv_plan* p, const float* w, const float* bias, void* stream) {
  if (!p || !w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_plan_update: null argument");
  return rgb_pack(p, w, bias, (hipStream_t)stream);
}

extern "C" void sntc_rgbconv_plan_destroy(sntc_rgbconv_plan* p) {
  if (p) rgb_free(p);
}

extern "C" int sntc_rgbconv_plan_set_workgroups(sntc_rgbconv_plan* p, int max_workgroups) {
  if (!p || max_workgroups < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_plan_set_workgroups: bad argument");
  p->max_workgroups = max_workgroups;
  return SNTC_OK;
}

extern "C" int64_t sntc_rgbconv_flops(const sntc_rgbconv_plan* p, int n, int h, int w) {
  if (!p || n < 0 || h < 0 || w < 0) return -1;
  const int64_t ho = (h + kStride - 1) / kStride, wo = (w + kStride - 1) / kStride;
  return 2 * (int64_t)n * ho * wo * p->k * p->k * kCin * p->cout;
}

extern "C" int sntc_rgbconv_forward(const sntc_rgbconv_plan* p, const float* x, int n, int h, int w, float* y, void* stream) {
  if (!p) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_forward: null plan");
  if (n < 0 || h < 0 || w < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_forward: negative size");
  if (n == 0 || h == 0 || w == 0) return SNTC_OK;
  if (!x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_forward: null argument");
  const int ho = (h + kStride - 1) / kStride, wo = (w + kStride - 1) / kStride;
  const int64_t xbytes = (int64_t)n * h * w * kCin * 4, ybytes = (int64_t)n * ho * wo * p->cout * 4;
  if (xbytes >= (1LL << 31) || ybytes >= (1LL << 31)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_forward: tensor of 2 GiB or more; split the batch");
  int cus = 0;
  if (int rc = rgb_init(&cus)) return rc;
  RGBArgs a{};
  a.x = x; a.y = y; a.wpack = p->wpack; a.bias = p->bias;
  a.xbytes = (unsigned)xbytes; a.ybytes = (unsigned)ybytes;
  a.N = n; a.H = h; a.W = w; a.Ho = ho; a.Wo = wo;
  a.k = p->k;
  if (p->kind == SNTC_SIGNAL_DOWN) {
    // tfc.SignalConv2D(corr=True, strides_down, "same_zeros"): the kernel is CENTRED, y[i] = sum_j w[j] x[s i + j - k / 2]
    a.pt = a.pl = p->k / 2;
  } else {
    // Keras SAME: pad_total = max((out - 1) s + k - in, 0), the smaller half in front
    a.pt = std::max((ho - 1) * kStride + p->k - h, 0) / 2;
    a.pl = std::max((wo - 1) * kStride + p->k - w, 0) / 2;
  }
  a.tiles_x = (wo + kTW - 1) / kTW;
  const int64_t nu = (int64_t)n * a.tiles_x * ho;
  if (nu >= (1LL << 31)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rgbconv_forward: too many units");
  a.nunits = (int)nu;
  a.act_m = p->act == SNTC_ACT_LEAKY_RELU ? 0.2f : 0.0f;
  a.act_c = p->act == SNTC_ACT_NONE ? -INFINITY : p->act == SNTC_ACT_RELU ? 0.0f : -0.0f;
  const int grid = (int)std::min<int64_t>((nu + kWaves - 1) / kWaves, p->max_workgroups > 0 ? p->max_workgroups : cus);
  hipStream_t s = (hipStream_t)stream;
#define RGB_LAUNCH(NT)                                                                                                   \
  do {                                                                                                                   \
    const bool act = p->act != SNTC_ACT_NONE;                                                                            \
    if (p->k == 5 && !act) hipLaunchKernelGGL((rgb_conv_kernel<NT, 5, false>), dim3(grid), dim3(512), RGBCfg<NT>::LDS, s, a);  \
    else if (p->k == 5) hipLaunchKernelGGL((rgb_conv_kernel<NT, 5, true>), dim3(grid), dim3(512), RGBCfg<NT>::LDS, s, a);      \
    else if (!act) hipLaunchKernelGGL((rgb_conv_kernel<NT, 0, false>), dim3(grid), dim3(512), RGBCfg<NT>::LDS, s, a);          \
    else hipLaunchKernelGGL((rgb_conv_kernel<NT, 0, true>), dim3(grid), dim3(512), RGBCfg<NT>::LDS, s, a);                     \
  } while (0)
  switch (p->cout) {
    case 128: RGB_LAUNCH(4); break;
    case 192: RGB_LAUNCH(6); break;
    default: RGB_LAUNCH(8); break;
  }
#undef RGB_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "first-layer launch");
  return SNTC_OK;
}

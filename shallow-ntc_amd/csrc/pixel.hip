// pixel.hip -- the HBM-bound ends of the path: reflect pad / crop, uint8 quantisation + squared
// error, small-channel GDN1, and the fused tail of the two-layer synthesis.
#include <algorithm>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ pad / crop
// y[n,hp,wp,c] <- x[n,h,w,c]; rows/cols >= h/w mirror without repeating the edge (tf.pad REFLECT).
__global__ void pad_reflect_kernel(const float* __restrict__ x, int h, int w, int c, int hp, int wp,
                                   float* __restrict__ y, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % c);
    int64_t t = i / c;
    const int j = (int)(t % wp);
    t /= wp;
    const int r = (int)(t % hp);
    const int64_t b = t / hp;
    const int sr = r < h ? r : 2 * h - 2 - r;
    const int sj = j < w ? j : 2 * w - 2 - j;
    y[i] = x[((b * h + sr) * w + sj) * c + k];
  }
}

// y[n,hp,wp,c] <- x[n,h,w,c] placed at (top, left), zeros elsewhere (the explicit form of a convolution's SAME padding).
// One block row per output image row: a row is one contiguous run of floats, shifted by left * c -- no per-element division.
__global__ void __launch_bounds__(256) pad_zero_kernel(const float* __restrict__ x, int h, int w, int c, int top, int left, int hp, int wp,
                                                       float* __restrict__ y) {
  const int64_t row = blockIdx.y;                      // (image, padded row)
  const int64_t b = row / hp;
  const int r = (int)(row - b * hp) - top;
  const bool row_ok = (unsigned)r < (unsigned)h;
  const int lo = left * c, hi = (left + w) * c, len = wp * c;
  const float* src = x + ((b * h + (row_ok ? r : 0)) * (int64_t)w) * c - lo;
  float* dst = y + row * (int64_t)len;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < len; j += gridDim.x * blockDim.x)
    dst[j] = (row_ok && j >= lo && j < hi) ? src[j] : 0.0f;
}

__global__ void crop_kernel(const float* __restrict__ x, int hp, int wp, int c, int h, int w, float* __restrict__ y,
                            int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % c);
    int64_t t = i / c;
    const int j = (int)(t % w);
    t /= w;
    const int r = (int)(t % h);
    const int64_t b = t / h;
    y[i] = x[((b * hp + r) * wp + j) * c + k];
  }
}

// ------------------------------------------------------------------ pixels + SSE
__device__ __forceinline__ int to_pixel(float v) {
  // data_lib.py:48-52 unnormalize_image, image_utils.py:22-23 tf.round (half-to-even) + saturate_cast
  const float p = rintf((v + 0.5f) * 255.0f);
  return (int)fminf(fmaxf(p, 0.0f), 255.0f);
}

// grid (blocks, n): one image per blockIdx.y; rows of w*c contiguous floats.
__global__ void __launch_bounds__(256) pixels_sse_kernel(const float* __restrict__ x, const float* __restrict__ xh, int h, int w,
                                                         int c, int hs, int ws, uint8_t* __restrict__ px,
                                                         unsigned long long* __restrict__ sse) {
  const int img = blockIdx.y;
  const int64_t per = (int64_t)h * w * c;
  const int rowlen = w * c;
  unsigned long long acc = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / rowlen);
    const int o = (int)(i - (int64_t)r * rowlen);
    const int b = to_pixel(xh[((int64_t)img * hs + r) * ws * c + o]);
    if (px) px[img * per + i] = (uint8_t)b;
    if (x) {
      const int d = to_pixel(x[img * per + i]) - b;
      acc += (unsigned)(d * d);
    }
  }
  if (!x) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse + img, part[0] + part[1] + part[2] + part[3]);
}

// The same four values per thread: 16-B loads, one packed 32-bit pixel store, 32-bit partial sums per iteration (4 x 255^2 fits
// with room), per row a 32-bit index split.  Needs rowlen % 4 == 0 and (ws * c) % 4 == 0, i.e. 16-B aligned rows in both tensors
// (every image width the configs use); the host falls back to the scalar kernel otherwise.  Same integers, same order-independent sum.
__global__ void __launch_bounds__(256) pixels_sse_vec4_kernel(const float* __restrict__ x, const float* __restrict__ xh, int h, int w,
                                                              int c, int hs, int ws, uint8_t* __restrict__ px,
                                                              unsigned long long* __restrict__ sse) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int img = blockIdx.y;
  const int64_t per = (int64_t)h * w * c;
  const unsigned row4 = (unsigned)(w * c) >> 2, srow4 = (unsigned)(ws * c) >> 2;
  const unsigned per4 = (unsigned)(per >> 2);
  const f32x4* xv = x ? reinterpret_cast<const f32x4*>(x + img * per) : nullptr;
  const f32x4* hv = reinterpret_cast<const f32x4*>(xh + (int64_t)img * hs * ws * c);
  unsigned* pv = px ? reinterpret_cast<unsigned*>(px + img * per) : nullptr;
  unsigned long long acc = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += gridDim.x * blockDim.x) {
    const unsigned r = i / row4;
    const f32x4 b4 = hv[r * srow4 + (i - r * row4)];
    const int b0 = to_pixel(b4[0]), b1 = to_pixel(b4[1]), b2 = to_pixel(b4[2]), b3 = to_pixel(b4[3]);
    if (pv) pv[i] = (unsigned)b0 | ((unsigned)b1 << 8) | ((unsigned)b2 << 16) | ((unsigned)b3 << 24);
    if (xv) {
      const f32x4 a4 = xv[i];
      const int d0 = to_pixel(a4[0]) - b0, d1 = to_pixel(a4[1]) - b1, d2 = to_pixel(a4[2]) - b2, d3 = to_pixel(a4[3]) - b3;
      acc += (unsigned)(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
    }
  }
  if (!x) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse + img, part[0] + part[1] + part[2] + part[3]);
}

__global__ void __launch_bounds__(256) float_sse_kernel(const float* __restrict__ x, const float* __restrict__ xh, int h, int w,
                                                        int c, int hs, int ws, double* __restrict__ sse) {
  const int img = blockIdx.y;
  const int64_t per = (int64_t)h * w * c;
  const int rowlen = w * c;
  double acc = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / rowlen);
    const int o = (int)(i - (int64_t)r * rowlen);
    const float a = (x[img * per + i] + 0.5f) * 255.0f;
    const float b = (xh[((int64_t)img * hs + r) * ws * c + o] + 0.5f) * 255.0f;
    const float d = a - b;
    acc += (double)(d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse + img, part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------ small-channel GDN
// One pixel per thread; beta/gamma in LDS.  norm_j = beta_j + sum_i pool(x_i) gamma[i][j].
template <int C>
__global__ void __launch_bounds__(256) gdn_small_kernel(const float* __restrict__ x, int64_t npix, const float* __restrict__ beta,
                                                        const float* __restrict__ gamma, int inverse, int alpha, int eps_half,
                                                        float* __restrict__ y) {
  __shared__ float sg[C * C];
  __shared__ float sb[C];
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) sg[i] = gamma[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) sb[i] = beta[i];
  __syncthreads();
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    float v[C], pool[C];
#pragma unroll
    for (int i = 0; i < C; i += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(x + p * C + i);
      v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
    }
#pragma unroll
    for (int i = 0; i < C; ++i) pool[i] = alpha == 1 ? fabsf(v[i]) : v[i] * v[i];
#pragma unroll
    for (int j0 = 0; j0 < C; j0 += 4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        float nrm = sb[j];
#pragma unroll
        for (int i = 0; i < C; ++i) nrm += pool[i] * sg[i * C + j];
        if (eps_half) nrm = sqrtf(nrm);
        o[e] = inverse ? v[j] * nrm : v[j] / nrm;
      }
      *reinterpret_cast<f32x4*>(y + p * C + j0) = o;
    }
  }
}

// ------------------------------------------------------------------ two-layer synthesis tail
// h = act(t[..., :CH]) (+ t[..., CH:]) at half resolution, then Conv2DTranspose 5x5 / 2 SAME (pt = 1)
// to 3 channels.  Block = 16 x 16 macro pixels q; thread (qy, qx) emits the 2x2 output quad
// oy = 2 qy + phi_y - 1, ox = 2 qx + phi_x - 1 (phi in {0,1}): phase 0 uses ky in {0,2,4} from rows
// qy, qy-1, qy-2; phase 1 uses ky in {1,3} from rows qy, qy-1.  The 18 x 18 x CH input tile of h
// lives in LDS, so the half-resolution activation is read from HBM once.
constexpr int tail_px(int ch) { return (ch / 4) % 2 ? ch : ch + 4; }              // 12 -> 12, 24 -> 28, 48 -> 52 words
constexpr int tail_row(int ch) { return (18 * tail_px(ch) + 63) / 64 * 64; }     // 256, 512, 960 words

template <int CH>
__global__ void __launch_bounds__(256) two_layer_tail_kernel(const float* __restrict__ t, int hh, int wh, int has_res,
                                                             int act_kind, const float* __restrict__ beta,
                                                             const float* __restrict__ gamma, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float* __restrict__ xhat,
                                                             int oh, int ow, const float* __restrict__ ref,
                                                             uint8_t* __restrict__ pix, unsigned long long* __restrict__ sse) {
  // xhat != NULL: float reconstruction [n, 2hh, 2wh, 3].  pix != NULL: the decoder's last steps fused in -- crop to oh x ow
  // (unpad_images), (v + .5) * 255, round half to even, saturate -> uint8 [n, oh, ow, 3]; with ref (float [n, oh, ow, 3])
  // also the per-image integer squared error of the two quantised images (mse_psnr), one u64 atomic per workgroup.
  constexpr int TQ = 16, TH = TQ + 2;
  // LDS image of the tile: pixel stride PX words with PX / 4 odd (so the 16 pixels of a row sit on 16 different 16-B bank
  // slots), row stride ROW a multiple of 64 words (so the second row a ds_read_b128 lane group touches -- lanes {0-3, 12-15}
  // read row ty, lanes {20-27} row ty + 1 -- lands on exactly the slots the first one leaves free): conflict-free fragment
  // reads (round 2 measured 6.1 M SQ_LDS_BANK_CONFLICT cycles per launch with the dense [18][18][CH] image)
  constexpr int PX = tail_px(CH), ROW = tail_row(CH);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sh = reinterpret_cast<float*>(smem);   // [TH][ROW], pixel lx of row ly at ly * ROW + lx * PX
  float* sg = sh + TH * ROW;                     // [CH][CH]
  float* sb = sg + CH * CH;                      // [CH]
  const int img = blockIdx.z;
  const int qy0 = blockIdx.y * TQ, qx0 = blockIdx.x * TQ;
  const int c2 = has_res ? 2 * CH : CH;
  const bool use_gdn = act_kind == 1 || act_kind == 2;
  if (use_gdn) {
    for (int i = threadIdx.x; i < CH * CH; i += 256) sg[i] = gamma[i];
    for (int i = threadIdx.x; i < CH; i += 256) sb[i] = beta[i];
  }
  __syncthreads();
  // stage 1: activation + residual add into the LDS tile (zeros outside the image)
  for (int p = threadIdx.x; p < TH * TH; p += 256) {
    const int ly = p / TH, lx = p - ly * TH;
    const int iy = qy0 - 2 + ly, ix = qx0 - 2 + lx;
    float* dst = sh + ly * ROW + lx * PX;
    if ((unsigned)iy < (unsigned)hh && (unsigned)ix < (unsigned)wh) {
      const float* src = t + (((int64_t)img * hh + iy) * wh + ix) * c2;
      float v[CH];
#pragma unroll
      for (int i = 0; i < CH; i += 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(src + i);
        v[i] = q[0]; v[i + 1] = q[1]; v[i + 2] = q[2]; v[i + 3] = q[3];
      }
#pragma unroll
      for (int j0 = 0; j0 < CH; j0 += 4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = j0 + e;
          float r = v[j];
          if (use_gdn) {
            float nrm = sb[j];
#pragma unroll
            for (int i = 0; i < CH; ++i) nrm += fabsf(v[i]) * sg[i * CH + j];
            r = act_kind == 1 ? v[j] * nrm : v[j] / nrm;
          } else if (act_kind == 3) {
            r = fmaxf(r, 0.0f);
          } else if (act_kind == 4) {
            r = r >= 0.0f ? r : 0.2f * r;
          }
          o[e] = r;
        }
        if (has_res) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(src + CH + j0);
          o = o + q;
        }
        *reinterpret_cast<f32x4*>(dst + j0) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < CH; i += 4) *reinterpret_cast<f32x4*>(dst + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  // stage 2
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int qy = qy0 + ty, qx = qx0 + tx;
  const int ho = 2 * hh, wo = 2 * wh;
  const float bias0 = b2[0], bias1 = b2[1], bias2 = b2[2];
  unsigned long long se = 0;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      const int oy = 2 * qy + py - 1, ox = 2 * qx + px - 1;
      float a0 = bias0, a1 = bias1, a2 = bias2;
#pragma unroll
      for (int jy = 0; jy < 3 - py; ++jy) {
#pragma unroll
        for (int jx = 0; jx < 3 - px; ++jx) {
          const int ky = py + 2 * jy, kx = px + 2 * jx;
          const float* hp = sh + (ty + 2 - jy) * ROW + (tx + 2 - jx) * PX;
          // compile-time offsets from a kernel-argument pointer: the weights arrive by scalar loads (SGPRs),
          // not through LDS -- 3/4 of the LDS reads of this loop were weight reads
          const float* wp = w2 + (ky * 5 + kx) * 3 * CH;
#pragma unroll
          for (int i = 0; i < CH; i += 4) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + i);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + i);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + CH + i);
            const f32x4 w2v = *reinterpret_cast<const f32x4*>(wp + 2 * CH + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a0 = fmaf(hv[e], w0[e], a0);
              a1 = fmaf(hv[e], w1[e], a1);
              a2 = fmaf(hv[e], w2v[e], a2);
            }
          }
        }
      }
      if (xhat && (unsigned)oy < (unsigned)ho && (unsigned)ox < (unsigned)wo) {
        float* dst = xhat + (((int64_t)img * ho + oy) * wo + ox) * 3;
        dst[0] = a0; dst[1] = a1; dst[2] = a2;
      }
      if (pix && (unsigned)oy < (unsigned)oh && (unsigned)ox < (unsigned)ow) {
        const int64_t o = (((int64_t)img * oh + oy) * ow + ox) * 3;
        const int q0 = to_pixel(a0), q1 = to_pixel(a1), q2 = to_pixel(a2);
        pix[o] = (uint8_t)q0; pix[o + 1] = (uint8_t)q1; pix[o + 2] = (uint8_t)q2;
        if (ref) {
          const int d0 = to_pixel(ref[o]) - q0, d1 = to_pixel(ref[o + 1]) - q1, d2 = to_pixel(ref[o + 2]) - q2;
          se += (unsigned)(d0 * d0 + d1 * d1 + d2 * d2);
        }
      }
    }
  }
  if (pix && ref) {                     // integer sum: exact and order-independent
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_down(se, o, 64);
    unsigned long long* part = reinterpret_cast<unsigned long long*>(sg);      // the gamma table is dead by now
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = se;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sse + img, part[0] + part[1] + part[2] + part[3]);
  }
}

}  // namespace sntc

using namespace sntc;

static int blocks_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int sntc_pad_reflect(const float* x, int n, int h, int w, int c, int hp, int wp, float* y, void* stream) {
  if (!x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pad_reflect: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || hp < h || wp < w || hp - h >= h || wp - w >= w)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_pad_reflect: reflect padding needs pad < size");
  const int64_t total = (int64_t)n * hp * wp * c;
  hipLaunchKernelGGL(pad_reflect_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, h, w, c, hp, wp, y, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_pad_zero(const float* x, int n, int h, int w, int c, int top, int left, int hp, int wp, float* y, void* stream) {
  if (!x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pad_zero: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || top < 0 || left < 0 || hp < h + top || wp < w + left)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_pad_zero: bad sizes");
  if ((int64_t)n * hp > 0x7fffffffLL / 1 || (int64_t)wp * c > 0x7fffffffLL) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pad_zero: tensor too large");
  const int bx = std::max(1, std::min(8, (wp * c + 1023) / 1024));
  hipLaunchKernelGGL(pad_zero_kernel, dim3(bx, (unsigned)(n * hp)), dim3(256), 0, (hipStream_t)stream, x, h, w, c, top, left, hp, wp, y);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_crop(const float* x, int n, int hp, int wp, int c, int h, int w, float* y, void* stream) {
  if (!x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_crop: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || hp < h || wp < w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_crop: bad sizes");
  const int64_t total = (int64_t)n * h * w * c;
  hipLaunchKernelGGL(crop_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, hp, wp, c, h, w, y, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

// y[p, :ca] = a[p, :], y[p, ca:ca+cb] = b[p, :] -- or, with b == NULL, a channel of ones followed by cb - 1 zero channels (the
// constant input channel of JPEGLikeSynthesis(use_offset=True), reference common/transforms.py:291-293, padded to a 16-channel slab).
__global__ void __launch_bounds__(256) concat_channels_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                                              float* __restrict__ y, int64_t total) {
  const int c = ca + cb;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c;
    const int k = (int)(i - p * c);
    y[i] = k < ca ? a[p * ca + k] : (b ? b[p * cb + (k - ca)] : (k == ca ? 1.0f : 0.0f));
  }
}

extern "C" int sntc_concat_channels(const float* a, int ca, const float* b, int cb, int64_t npix, float* y, void* stream) {
  if (!a || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_concat_channels: null argument");
  if (npix < 1 || ca < 1 || cb < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_concat_channels: bad sizes");
  const int64_t total = npix * (ca + cb);
  hipLaunchKernelGGL(concat_channels_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, a, ca, b, cb, y, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

// tf.nn.depth_to_space(x, block) in NHWC (DCR order: input channel = (dy * block + dx) * C_out + c): the upsampling step of
// TwoLayerResSynthesis(res_type="d2s"), reference common/transforms.py:341-348.  One thread per output element.
__global__ void __launch_bounds__(256) depth_to_space_kernel(const float* __restrict__ x, int h, int w, int c, int bs,
                                                             float* __restrict__ y, int64_t total) {
  const int co = c / (bs * bs);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % co);
    int64_t r = i / co;
    const int ox = (int)(r % ((int64_t)w * bs));
    r /= (int64_t)w * bs;
    const int oy = (int)(r % ((int64_t)h * bs));
    const int64_t n = r / ((int64_t)h * bs);
    const int iy = oy / bs, dy = oy - iy * bs, ix = ox / bs, dx = ox - ix * bs;
    y[i] = x[((n * h + iy) * w + ix) * c + (dy * bs + dx) * co + ch];
  }
}

extern "C" int sntc_depth_to_space(const float* x, int n, int h, int w, int c, int block, float* y, void* stream) {
  if (!x || !y) return fail(SNTC_ERR_BAD_SHAPE, "sntc_depth_to_space: null argument");
  if (n < 1 || h < 1 || w < 1 || block < 1 || c < 1 || c % (block * block))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_depth_to_space: channels must be a multiple of block^2");
  const int64_t total = (int64_t)n * h * w * c;
  hipLaunchKernelGGL(depth_to_space_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, h, w, c, block, y, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_pixels_sse(const float* x, const float* x_hat, int n, int h, int w, int c, int hs, int ws,
                               uint8_t* pixels_out, unsigned long long* sse_out, void* stream) {
  if (!x_hat || (x && !sse_out) || (!x && !pixels_out)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pixels_sse: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || hs < h || ws < w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pixels_sse: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (x) if (int zrc = zero_async(sse_out, sizeof(unsigned long long) * n, s)) return zrc;
  const int64_t per = (int64_t)h * w * c;
  const bool vec = (w * c) % 4 == 0 && (ws * c) % 4 == 0 && per < (1LL << 33) && ((int64_t)hs * ws * c) % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x_hat)) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(pixels_out) % 4 == 0;
  if (vec) {
    int b = blocks_for(per >> 2);
    // with an SSE every block ends in ONE atomic on its image's word, and atomics on one address serialise at the memory side
    // (~0.1 us each: 1024 blocks per image cost 100 us by themselves): about 2048 blocks per launch, at most 256 per image
    const int cap = x ? std::max(16, std::min(256, 2048 / n)) : 1024;
    if (b > cap) b = cap;
    hipLaunchKernelGGL(pixels_sse_vec4_kernel, dim3(b, n), dim3(256), 0, s, x, x_hat, h, w, c, hs, ws, pixels_out, sse_out);
    SNTC_HIP(hipGetLastError());
    return SNTC_OK;
  }
  int b = blocks_for(per);
  const int cap1 = x ? std::max(16, std::min(256, 2048 / n)) : 512;
  if (b > cap1) b = cap1;
  hipLaunchKernelGGL(pixels_sse_kernel, dim3(b, n), dim3(256), 0, s, x, x_hat, h, w, c, hs, ws, pixels_out, sse_out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_float_sse(const float* x, const float* x_hat, int n, int h, int w, int c, int hs, int ws,
                              double* sse_out, void* stream) {
  if (!x || !x_hat || !sse_out) return fail(SNTC_ERR_BAD_SHAPE, "sntc_float_sse: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || hs < h || ws < w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_float_sse: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(sse_out, sizeof(double) * n, s)) return zrc;
  const int64_t per = (int64_t)h * w * c;
  int b = blocks_for(per);
  if (b > 512) b = 512;
  hipLaunchKernelGGL(float_sse_kernel, dim3(b, n), dim3(256), 0, s, x, x_hat, h, w, c, hs, ws, sse_out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

template <int C>
static int launch_gdn(const float* x, int64_t npix, const float* beta, const float* gamma, int inverse, int alpha,
                      int eps_half, float* y, hipStream_t s) {
  hipLaunchKernelGGL((gdn_small_kernel<C>), dim3(blocks_for(npix)), dim3(256), 0, s, x, npix, beta, gamma, inverse, alpha,
                     eps_half, y);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gdn_small(const float* x, int64_t npix, int c, const float* beta, const float* gamma, int inverse,
                              int alpha, int epsilon_is_half, float* y, void* stream) {
  if (!x || !beta || !gamma || !y || npix < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gdn_small: null argument");
  if (alpha != 1 && alpha != 2) return fail(SNTC_ERR_UNSUPPORTED, "sntc_gdn_small: alpha must be 1 or 2");
  hipStream_t s = (hipStream_t)stream;
  switch (c) {
    case 4: return launch_gdn<4>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    case 8: return launch_gdn<8>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    case 12: return launch_gdn<12>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    case 16: return launch_gdn<16>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    case 24: return launch_gdn<24>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    case 32: return launch_gdn<32>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    case 48: return launch_gdn<48>(x, npix, beta, gamma, inverse, alpha, epsilon_is_half, y, s);
    default: return fail(SNTC_ERR_UNSUPPORTED, "sntc_gdn_small: channels must be one of 4,8,12,16,24,32,48");
  }
}

template <int CH>
static int launch_tail(const float* t, int n, int hh, int wh, int has_res, int act_kind, const float* beta,
                       const float* gamma, const float* w2, const float* b2, float* x_hat, int oh, int ow, const float* ref,
                       uint8_t* px, unsigned long long* sse, hipStream_t s) {
  const size_t lds = sizeof(float) * (18 * tail_row(CH) + CH * CH + CH);
  static thread_local int attr_dev = -1;
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (attr_dev != dev) {
    SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&two_layer_tail_kernel<CH>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_dev = dev;
  }
  dim3 grid((wh + 1 + 15) / 16, (hh + 1 + 15) / 16, n);
  if (px && ref && sse) if (int zrc = zero_async(sse, sizeof(unsigned long long) * n, s)) return zrc;
  hipLaunchKernelGGL((two_layer_tail_kernel<CH>), grid, dim3(256), lds, s, t, hh, wh, has_res, act_kind, beta, gamma, w2,
                     b2, x_hat, oh, ow, ref, px, sse);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

static int tail_dispatch(const float* t, int n, int hh, int wh, int ch, int has_res, int act_kind, const float* beta,
                         const float* gamma, const float* w2, const float* b2, int k2, int s2, int cout, float* x_hat, int oh,
                         int ow, const float* ref, uint8_t* px, unsigned long long* sse, void* stream) {
  if (!t || !w2 || !b2 || (!x_hat && !px)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail: null argument");
  if (n < 1 || hh < 1 || wh < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail: empty input");
  if (k2 != 5 || s2 != 2 || cout != 3)
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_tail: only the 5x5 / stride-2 / 3-channel output layer is fused");
  if (act_kind < 0 || act_kind > 4) return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_tail: unknown activation");
  if ((act_kind == 1 || act_kind == 2) && (!beta || !gamma))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail: GDN parameters missing");
  if (n > 65535) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail: batch too large");
  if (px && (oh < 1 || ow < 1 || oh > 2 * hh || ow > 2 * wh))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_pixels: crop must lie inside the reconstruction");
  if (ref && !sse) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_pixels: a reference needs an sse output");
  hipStream_t s = (hipStream_t)stream;
  switch (ch) {
    case 12: return launch_tail<12>(t, n, hh, wh, has_res, act_kind, beta, gamma, w2, b2, x_hat, oh, ow, ref, px, sse, s);
    case 24: return launch_tail<24>(t, n, hh, wh, has_res, act_kind, beta, gamma, w2, b2, x_hat, oh, ow, ref, px, sse, s);
    case 48: return launch_tail<48>(t, n, hh, wh, has_res, act_kind, beta, gamma, w2, b2, x_hat, oh, ow, ref, px, sse, s);
    default: return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_tail: hidden channels must be 12, 24 or 48");
  }
}

extern "C" int sntc_two_layer_tail(const float* t, int n, int hh, int wh, int ch, int has_res, int act_kind,
                                   const float* beta, const float* gamma, const float* w2, const float* b2, int k2,
                                   int s2, int cout, float* x_hat, void* stream) {
  if (!x_hat) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail: null argument");
  return tail_dispatch(t, n, hh, wh, ch, has_res, act_kind, beta, gamma, w2, b2, k2, s2, cout, x_hat, 0, 0, nullptr, nullptr,
                       nullptr, stream);
}

extern "C" int sntc_two_layer_tail_pixels(const float* t, int n, int hh, int wh, int ch, int has_res, int act_kind,
                                          const float* beta, const float* gamma, const float* w2, const float* b2, int k2,
                                          int s2, int cout, int h, int w, const float* ref, uint8_t* pixels, uint64_t* sse,
                                          void* stream) {
  if (!pixels) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_pixels: null argument");
  return tail_dispatch(t, n, hh, wh, ch, has_res, act_kind, beta, gamma, w2, b2, k2, s2, cout, nullptr, h, w, ref, pixels,
                       reinterpret_cast<unsigned long long*>(sse), stream);
}

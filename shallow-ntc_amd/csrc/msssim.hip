// msssim.hip -- SSIM / MS-SSIM statistics (reference mshyper/models.py:321-336 ->
// tf.image.ssim / tf.image.ssim_multiscale).  HBM-bound: every scale reads both images once.
//
// One block = a 16 x 16 tile of filter outputs of one image; the 26 x 26 x C input tiles of both images
// sit in LDS and each thread evaluates the 11 x 11 Gaussian moments (mu_x, mu_y, E[xy], E[x^2 + y^2]) of
// its pixel for every channel, then luminance and contrast-structure; per-(image, channel) sums are
// reduced with wave shuffles + one double atomic per block.  The host finishes the (tiny) geometric mean.
#include <cmath>
#include "sntc_internal.h"

namespace sntc {

constexpr int kWin = 11;
constexpr int kTile = 16;
constexpr int kIn = kTile + kWin - 1;   // 26

struct GaussWin {
  float w[kWin];
};

template <int C>
__global__ void __launch_bounds__(256) ssim_scale_kernel(const float* __restrict__ a, const float* __restrict__ b, int h, int w,
                                                         GaussWin win, float c1, float c2, double* __restrict__ ssim_sum,
                                                         double* __restrict__ cs_sum) {
  __shared__ float sa[kIn * kIn * C];
  __shared__ float sb[kIn * kIn * C];
  const int img = blockIdx.z;
  const int oy0 = blockIdx.y * kTile, ox0 = blockIdx.x * kTile;
  const int ho = h - kWin + 1, wo = w - kWin + 1;
  const float* ab = a + (size_t)img * h * w * C;
  const float* bb = b + (size_t)img * h * w * C;
  for (int i = threadIdx.x; i < kIn * kIn * C; i += 256) {
    const int ch = i % C;
    const int p = i / C;
    const int ly = p / kIn, lx = p - ly * kIn;
    const int iy = min(oy0 + ly, h - 1), ix = min(ox0 + lx, w - 1);     // clamped reads feed masked outputs only
    sa[i] = ab[((size_t)iy * w + ix) * C + ch];
    sb[i] = bb[((size_t)iy * w + ix) * C + ch];
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const bool valid = (oy0 + ty) < ho && (ox0 + tx) < wo;
  float s_out[C], cs_out[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    float m0 = 0.f, m1 = 0.f, exy = 0.f, esq = 0.f;
    for (int i = 0; i < kWin; ++i) {
      float r0 = 0.f, r1 = 0.f, rxy = 0.f, rsq = 0.f;
#pragma unroll
      for (int j = 0; j < kWin; ++j) {
        const int idx = ((ty + i) * kIn + tx + j) * C + ch;
        const float x = sa[idx], y = sb[idx], wj = win.w[j];
        r0 += wj * x;
        r1 += wj * y;
        rxy += wj * (x * y);
        rsq += wj * (x * x + y * y);
      }
      const float wi = win.w[i];
      m0 += wi * r0;
      m1 += wi * r1;
      exy += wi * rxy;
      esq += wi * rsq;
    }
    const float num0 = 2.0f * m0 * m1, den0 = m0 * m0 + m1 * m1;
    const float lum = (num0 + c1) / (den0 + c1);
    const float cs = (2.0f * exy - num0 + c2) / (esq - den0 + c2);
    s_out[ch] = valid ? lum * cs : 0.0f;
    cs_out[ch] = valid ? cs : 0.0f;
  }
  __shared__ double part[4][2 * C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    double v0 = (double)s_out[ch], v1 = (double)cs_out[ch];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      v0 += __shfl_down(v0, o, 64);
      v1 += __shfl_down(v1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      part[threadIdx.x >> 6][2 * ch] = v0;
      part[threadIdx.x >> 6][2 * ch + 1] = v1;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const double s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    const int ch = threadIdx.x >> 1;
    atomicAdd(((threadIdx.x & 1) ? cs_sum : ssim_sum) + (size_t)img * C + ch, s);
  }
}

// 2 x 2 average pooling; an odd size is first extended by repeating its last row / column (tf.pad SYMMETRIC)
__global__ void avgpool2_kernel(const float* __restrict__ x, int h, int w, int c, float* __restrict__ y, int64_t total) {
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % c);
    int64_t t = i / c;
    const int ox = (int)(t % wo);
    t /= wo;
    const int oy = (int)(t % ho);
    const int64_t b = t / ho;
    const int y0 = 2 * oy, y1 = min(2 * oy + 1, h - 1), x0 = 2 * ox, x1 = min(2 * ox + 1, w - 1);
    const float* p = x + b * h * w * c;
    y[i] = 0.25f * (p[((int64_t)y0 * w + x0) * c + k] + p[((int64_t)y0 * w + x1) * c + k] +
                    p[((int64_t)y1 * w + x0) * c + k] + p[((int64_t)y1 * w + x1) * c + k]);
  }
}

// crop + (v + .5) * 255 + round-half-even + clamp, kept as float (the SSIM input)
__global__ void pixels_float_kernel(const float* __restrict__ xh, int h, int w, int c, int hs, int ws, float* __restrict__ out,
                                    int64_t total) {
  const int rowlen = w * c;
  const int64_t per = (int64_t)h * rowlen;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per;
    const int64_t rem = i - b * per;
    const int r = (int)(rem / rowlen), o = (int)(rem - (int64_t)r * rowlen);
    const float v = xh[(b * hs + r) * ws * c + o];
    out[i] = fminf(fmaxf(rintf((v + 0.5f) * 255.0f), 0.0f), 255.0f);
  }
}

}  // namespace sntc

using namespace sntc;

static int blocks_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int sntc_ssim_scale(const float* a, const float* b, int n, int h, int w, int c, float max_val, double* ssim_sum,
                               double* cs_sum, void* stream) {
  if (!a || !b || !ssim_sum || !cs_sum) return fail(SNTC_ERR_BAD_SHAPE, "sntc_ssim_scale: null argument");
  if (n < 1 || n > 65535 || h < kWin || w < kWin) return fail(SNTC_ERR_BAD_SHAPE, "sntc_ssim_scale: image smaller than the 11 x 11 window");
  GaussWin win;
  double s = 0, g[kWin];
  for (int i = 0; i < kWin; ++i) {
    const double d = i - (kWin - 1) / 2.0;
    g[i] = std::exp(-0.5 * d * d / (1.5 * 1.5));
    s += g[i];
  }
  for (int i = 0; i < kWin; ++i) win.w[i] = (float)(g[i] / s);
  const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
  hipStream_t st = (hipStream_t)stream;
  if (int zrc = zero_async(ssim_sum, sizeof(double) * n * c, st)) return zrc;
  if (int zrc = zero_async(cs_sum, sizeof(double) * n * c, st)) return zrc;
  const int ho = h - kWin + 1, wo = w - kWin + 1;
  dim3 grid((wo + kTile - 1) / kTile, (ho + kTile - 1) / kTile, n);
  switch (c) {
    case 1: hipLaunchKernelGGL((ssim_scale_kernel<1>), grid, dim3(256), 0, st, a, b, h, w, win, c1, c2, ssim_sum, cs_sum); break;
    case 3: hipLaunchKernelGGL((ssim_scale_kernel<3>), grid, dim3(256), 0, st, a, b, h, w, win, c1, c2, ssim_sum, cs_sum); break;
    default: return fail(SNTC_ERR_UNSUPPORTED, "sntc_ssim_scale: 1 or 3 channels");
  }
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_avgpool2_symmetric(const float* x, int n, int h, int w, int c, float* y, void* stream) {
  if (!x || !y || n < 1 || h < 1 || w < 1 || c < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_avgpool2_symmetric: bad argument");
  const int64_t total = (int64_t)n * ((h + 1) / 2) * ((w + 1) / 2) * c;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, h, w, c, y, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_pixels_float(const float* x_hat, int n, int h, int w, int c, int hs, int ws, float* out, void* stream) {
  if (!x_hat || !out) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pixels_float: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || hs < h || ws < w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_pixels_float: bad sizes");
  const int64_t total = (int64_t)n * h * w * c;
  hipLaunchKernelGGL(pixels_float_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x_hat, h, w, c, hs, ws, out, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

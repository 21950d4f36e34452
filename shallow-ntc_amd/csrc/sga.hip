// sga.hip -- element-wise kernels of SGA iterative inference:
//   stochastic Gumbel annealing sample + its derivative, rate terms and their gradients for both
//   entropy models, distortion gradient, two-layer-tail backward, fused Adam on the latents.
// The contractions of the backward pass (input gradients of the transposed convolutions) reuse the
// gather-GEMM: the adjoint of Conv2DTranspose(k, s, SAME) with kernel W[kh,kw,Cout,Cin] is
// Conv2D(k, s, SAME) with the same array read as HWIO.
//
// Reference: common/latent_rvs_utils.py:8-48 (sga_round), mshyper/models.py:260-268,285-291,343,
// 397-408 (loss terms and the variables that receive gradients), common/data_lib.py:48-52.
#include <cmath>
#include <algorithm>
#include "device_math.h"

namespace sntc {


constexpr float kSgaEps = 1e-5f;           // latent_rvs_utils.py:9 epsilon

// ---- counter-based uniform -> Gumbel (statistical parity with tfp's RelaxedOneHotCategorical) ----
__device__ __forceinline__ float gumbel_from(unsigned long long seed, unsigned long long step, unsigned long long idx, int k) {
  return -logf(-logf(uniform01(splitmix64(stream_key(seed, step) + 2 * idx + k))));
}

// out = w0 floor(mu) + w1 ceil(mu), w = softmax((logits + g) / tau); also d out / d mu.
__device__ __forceinline__ void sga_sample(float mu, float tau, float g0, float g1, float* out, float* dout) {
  const float fl = floorf(mu), ce = ceilf(mu);
  const float a = fminf(fmaxf(mu - fl, -1.0f + kSgaEps), 1.0f - kSgaEps);
  const float b = fminf(fmaxf(ce - mu, -1.0f + kSgaEps), 1.0f - kSgaEps);
  const float l0 = -atanhf(a) / tau, l1 = -atanhf(b) / tau;
  const float d = ((l1 + g1) - (l0 + g0)) / tau;
  const float w1 = 1.0f / (1.0f + expf(-d));
  *out = (1.0f - w1) * fl + w1 * ce;
  // d l0 / d mu = -1/(tau (1-a^2)) unless clipped; d l1 / d mu = +1/(tau (1-b^2)) unless clipped
  const bool a_free = (mu - fl) < 1.0f - kSgaEps, b_free = (ce - mu) < 1.0f - kSgaEps;
  const float dl0 = a_free ? -1.0f / (tau * (1.0f - a * a)) : 0.0f;
  const float dl1 = b_free ? 1.0f / (tau * (1.0f - b * b)) : 0.0f;
  *dout = (ce - fl) * w1 * (1.0f - w1) * (dl1 - dl0) / tau;
}

// Gradient through idx = clamp(exp(raw), 0, 63): tfc's _normalize_indexes bounds the indexes with math_ops.lower_bound /
// upper_bound, whose default gradient is "identity_if_towards" [DEP]: a saturated index still receives the gradient when a
// descent step would move it back towards the bound (d loss / d idx > 0 at the upper bound; the rate's weight in the loss
// is positive, so the sign of d bits / d sigma decides).  exp(raw) > 0, so the lower bound never binds.
__device__ __forceinline__ bool scale_index_gate(float e, float dbits_dsigma) {
  return e <= 63.0f || dbits_dsigma > 0.0f;
}

// ---- normal (y) : forward sample + rate + partial derivatives ----
// bits(v, sigma) = -log2 [Phi((v+.5)/s) - Phi((v-.5)/s)];  d bits/d v, d bits/d raw (through
// idx = clamp(exp(raw), 0, 63), sigma = exp(c0 + c1 idx)).
__global__ void __launch_bounds__(256) sga_normal_fwd_kernel(const float* __restrict__ y_loc, const float* __restrict__ hyper,
                                                             int64_t hw, int c, float tau, const float* __restrict__ noise,
                                                             unsigned long long seed, unsigned long long step,
                                                             float* __restrict__ y_tilde, float* __restrict__ sprime,
                                                             float* __restrict__ dbits_dv, float* __restrict__ dbits_draw,
                                                             double* __restrict__ bits) {
  const int img = blockIdx.y;
  const int64_t per = hw * c;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c;
    const int ch = (int)(i - p * c);
    const int64_t gi = img * per + i;
    const float mu = hyper[(img * hw + p) * 2 * c + ch];
    const float raw = hyper[(img * hw + p) * 2 * c + c + ch];
    const float g0 = noise ? noise[2 * gi] : gumbel_from(seed, step, (unsigned long long)gi, 0);
    const float g1 = noise ? noise[2 * gi + 1] : gumbel_from(seed, step, (unsigned long long)gi, 1);
    float v, sp;
    sga_sample(y_loc[gi] - mu, tau, g0, g1, &v, &sp);
    const float e = expf(raw);
    const float idx = fminf(fmaxf(e, 0.0f), 63.0f);
    const float sigma = expf(kLogScaleMin + kScaleFactor * idx);
    const float hi = (v + 0.5f) / sigma, lo = (v - 0.5f) / sigma;
    const bool right = hi > 0.0f;
    const float big = log_ndtr_f(right ? -lo : hi), small = log_ndtr_f(right ? -hi : lo);
    const float logp = big + log1pf(-expf(small - big));
    // phi(x)/p in the log domain
    const float lphi_hi = -0.5f * hi * hi - 0.91893853320467274f - logp;
    const float lphi_lo = -0.5f * lo * lo - 0.91893853320467274f - logp;
    const float r_hi = expf(lphi_hi), r_lo = expf(lphi_lo);
    const float dlogp_dv = (r_hi - r_lo) / sigma;
    const float dlogp_ds = -(r_hi * hi - r_lo * lo) / sigma;
    const float dsig_draw = scale_index_gate(e, -dlogp_ds) ? sigma * kScaleFactor * e : 0.0f;
    y_tilde[gi] = v + mu;
    sprime[gi] = sp;
    dbits_dv[gi] = -dlogp_dv * kInvLn2;
    dbits_draw[gi] = -dlogp_ds * dsig_draw * kInvLn2;
    acc += (double)(-logp * kInvLn2);
  }
  block_sum_to(acc, bits + img);
}

// g_yloc = (g_yt + w dbits_dv) s';  g_mu = g_yt (1 - s') - w dbits_dv s';  g_raw = w dbits_draw
__global__ void __launch_bounds__(256) sga_normal_bwd_kernel(const float* __restrict__ g_yt, const float* __restrict__ sprime,
                                                             const float* __restrict__ dbits_dv, const float* __restrict__ dbits_draw,
                                                             float w, int64_t npix, int c, float* __restrict__ g_yloc,
                                                             float* __restrict__ g_hyper) {
  const int64_t total = npix * c;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c;
    const int ch = (int)(i - p * c);
    const float g = g_yt[i], sp = sprime ? sprime[i] : 1.0f, dv = w * dbits_dv[i];
    g_yloc[i] = (g + dv) * sp;
    g_hyper[p * 2 * c + ch] = g * (1.0f - sp) - dv * sp;
    g_hyper[p * 2 * c + c + ch] = w * dbits_draw[i];
  }
}

// ---- deep factorized (z) ----
// logits L(x) and dL/dx
__device__ __forceinline__ void df_logits_grad(const float* __restrict__ rec, const DFDesc& d, float x, float* L, float* dL) {
  float hv[kMaxW] = {x, 0.f, 0.f, 0.f}, hd[kMaxW] = {1.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kMaxL; ++k) {
    if (k < d.nl) {
      const int fi = d.w[k], fo = d.w[k + 1];
      float nv[kMaxW] = {0.f, 0.f, 0.f, 0.f}, nd[kMaxW] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < kMaxW; ++o) {
        if (o < fo) {
          float s = rec[d.off_b[k] + o], ds = 0.0f;
#pragma unroll
          for (int i = 0; i < kMaxW; ++i)
            if (i < fi) {
              const float m = rec[d.off_m[k] + o * fi + i];
              s += m * hv[i];
              ds += m * hd[i];
            }
          if (k < d.nl - 1) {
            const float f = rec[d.off_f[k] + o], th = tanhf(s);
            ds *= 1.0f + f * (1.0f - th * th);
            s += f * th;
          }
          nv[o] = s;
          nd[o] = ds;
        }
      }
#pragma unroll
      for (int o = 0; o < kMaxW; ++o) { hv[o] = nv[o]; hd[o] = nd[o]; }
    }
  }
  *L = hv[0];
  *dL = hd[0];
}


__global__ void __launch_bounds__(256) sga_factorized_fwd_kernel(const float* __restrict__ rec_all, DFDesc d,
                                                                 const float* __restrict__ z_loc, int64_t hw, int c, float tau,
                                                                 const float* __restrict__ noise, unsigned long long seed,
                                                                 unsigned long long step, float* __restrict__ z_tilde,
                                                                 float* __restrict__ sprime, float* __restrict__ dbits_dz,
                                                                 double* __restrict__ bits) {
  const int img = blockIdx.y;
  const int64_t per = hw * c;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const int64_t gi = img * per + i;
    const float g0 = noise ? noise[2 * gi] : gumbel_from(seed ^ 0xA5A5A5A5ull, step, (unsigned long long)gi, 0);
    const float g1 = noise ? noise[2 * gi + 1] : gumbel_from(seed ^ 0xA5A5A5A5ull, step, (unsigned long long)gi, 1);
    float v, sp;
    sga_sample(z_loc[gi], tau, g0, g1, &v, &sp);
    const float* rec = rec_all + (size_t)ch * d.stride;
    float hi, dhi, lo, dlo;
    df_logits_grad(rec, d, v + 0.5f, &hi, &dhi);
    df_logits_grad(rec, d, v - 0.5f, &lo, &dlo);
    const bool right = hi > 0.0f;
    const float big = log_sigmoid_f(right ? -lo : hi), small = log_sigmoid_f(right ? -hi : lo);
    const float ratio = expf(small - big);                      // in [0, 1)
    const float logp = big + log1pf(-ratio);
    // dp/dv = s(hi) s(-hi) L'hi - s(lo) s(-lo) L'lo, divided by exp(big)
    float num;
    if (!right) num = sigmoid_f(-hi) * dhi - ratio * sigmoid_f(-lo) * dlo;   // / s(hi)
    else num = ratio * sigmoid_f(hi) * dhi - sigmoid_f(lo) * dlo;           // / s(-lo); equals -(...)? see below
    // right branch: p = s(-lo) - s(-hi); dp/dv = s(hi)s(-hi)L'hi - s(lo)s(-lo)L'lo; / s(-lo):
    //   = [s(-hi)/s(-lo)] s(hi) L'hi - s(lo) L'lo = ratio s(hi) L'hi - s(lo) L'lo  (matches num above)
    const float dlogp = num / (1.0f - ratio);
    z_tilde[gi] = v;
    sprime[gi] = sp;
    dbits_dz[gi] = -dlogp * kInvLn2;
    acc += (double)(-logp * kInvLn2);
  }
  block_sum_to(acc, bits + img);
}

// out = (g + w d) * s'
__global__ void __launch_bounds__(256) sga_chain_kernel(const float* __restrict__ g, const float* __restrict__ dbits,
                                                        const float* __restrict__ sprime, float w, int64_t total,
                                                        float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (g[i] + w * dbits[i]) * sprime[i];
}

// ---- UQLatentRV.sample / .quantize (reference common/latent_rvs_lib.py:77-116) as one element-wise kernel ----
// u = loc - offset (offset: NULL = 0, or a tensor read with a pixel stride: the mean half of the hyper-synthesis output);
// mode 0: round-half-even(u) (training=False, :95-102 / tfc.round_st forward); 1: loc + U(-.5, .5) ('unoise', :104-107: the
// offset plays no role); 2: sga_round (common/latent_rvs_utils.py:8-48); 3: tfc.soft_round(u, alpha) (:111-114) [DEP]:
// m = floor(u) + .5, out = m + tanh(alpha (u - m)) / (2 tanh(alpha / 2)), alpha bounded below by 1e-3 (identity under it).
__global__ void __launch_bounds__(256) uq_sample_kernel(const float* __restrict__ loc, const float* __restrict__ offset,
                                                        int64_t npix, int c, int ostride, int mode, float param,
                                                        const float* __restrict__ noise, unsigned long long seed,
                                                        unsigned long long step, float* __restrict__ out) {
  const int64_t total = npix * c;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c;
    const int ch = (int)(i - p * c);
    const float off = offset ? offset[p * ostride + ch] : 0.0f;
    const float x = loc[i];
    const float u = x - off;
    float v;
    if (mode == 0) {
      v = rintf(u) + off;
    } else if (mode == 1) {
      v = x + (noise ? noise[i] : uniform01(splitmix64(stream_key(seed, step) + (unsigned long long)i)) - 0.5f);
    } else if (mode == 2) {
      const float g0 = noise ? noise[2 * i] : gumbel_from(seed, step, (unsigned long long)i, 0);
      const float g1 = noise ? noise[2 * i + 1] : gumbel_from(seed, step, (unsigned long long)i, 1);
      float sp;
      sga_sample(u, param, g0, g1, &v, &sp);
      v += off;
    } else {
      if (param < 1e-3f) {
        v = x;
      } else {
        const float m = floorf(u) + 0.5f;
        v = m + tanhf(param * (u - m)) / (2.0f * tanhf(0.5f * param)) + off;
      }
    }
    out[i] = v;
  }
}

// ---- distortion: sse[n] of 255 (x - x_hat) and g_xhat = scale (x_hat - x) (zeros in the padded margin) ----
__global__ void __launch_bounds__(256) distortion_grad_kernel(const float* __restrict__ x, const float* __restrict__ xh, int h, int w,
                                                              int c, int hs, int ws, float scale, float* __restrict__ g,
                                                              double* __restrict__ sse) {
  const int img = blockIdx.y;
  const int64_t per_s = (int64_t)hs * ws * c;
  const int rowlen_s = ws * c;
  double acc = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per_s; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / rowlen_s);
    const int o = (int)(i - (int64_t)r * rowlen_s);
    float gv = 0.0f;
    if (r < h && o < w * c) {
      const float d = xh[img * per_s + i] - x[((int64_t)img * h + r) * w * c + o];
      gv = scale * d;
      const float d255 = 255.0f * d;
      acc += (double)(d255 * d255);
    }
    g[img * per_s + i] = gv;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse + img, part[0] + part[1] + part[2] + part[3]);
}

// ---- two-layer tail backward: g_t[..., :CH] = d act(base) (g_h), g_t[..., CH:2CH] = g_h (residual), zero padding to CP ----
// IGDN1: y_j = x_j n_j, n_j = beta_j + sum_i |x_i| gamma_ij  =>  dx_i = g_i n_i + sign(x_i) sum_j g_j x_j gamma_ij
// GDN1 : y_j = x_j / n_j                                         =>  dx_i = g_i / n_i - sign(x_i) sum_j g_j x_j gamma_ij / n_j^2
template <int CH>
__global__ void __launch_bounds__(256) tail_bwd_kernel(const float* __restrict__ t, const float* __restrict__ gh, int64_t npix,
                                                       int has_res, int act_kind, const float* __restrict__ beta,
                                                       const float* __restrict__ gamma, int cp, float* __restrict__ gt,
                                                       float* __restrict__ absx, float* __restrict__ gx) {
  __shared__ float sg[CH * CH];
  __shared__ float sb[CH];
  const bool use_gdn = act_kind == 1 || act_kind == 2;
  if (use_gdn) {
    for (int i = threadIdx.x; i < CH * CH; i += blockDim.x) sg[i] = gamma[i];
    for (int i = threadIdx.x; i < CH; i += blockDim.x) sb[i] = beta[i];
  }
  __syncthreads();
  const int c2 = has_res ? 2 * CH : CH;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    float xv[CH], g[CH];
#pragma unroll
    for (int i = 0; i < CH; i += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(t + p * c2 + i);
      const f32x4 b = *reinterpret_cast<const f32x4*>(gh + p * CH + i);
      xv[i] = a[0]; xv[i + 1] = a[1]; xv[i + 2] = a[2]; xv[i + 3] = a[3];
      g[i] = b[0]; g[i + 1] = b[1]; g[i + 2] = b[2]; g[i + 3] = b[3];
    }
    float* dst = gt + p * cp;
    if (absx) {                                             // operands of the IGDN1 parameter gradients (train step):
#pragma unroll                                              //   d gamma = |x|^T (g x) over pixels, d beta = column sums of g x
      for (int i = 0; i < CH; ++i) {
        absx[p * CH + i] = fabsf(xv[i]);
        gx[p * CH + i] = g[i] * xv[i];
      }
    }
    if (use_gdn) {
      float u[CH];   // per-j factor multiplying gamma_ij
      float nrm[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float n = sb[j];
#pragma unroll
        for (int i = 0; i < CH; ++i) n += fabsf(xv[i]) * sg[i * CH + j];
        nrm[j] = n;
        u[j] = act_kind == 1 ? g[j] * xv[j] : -g[j] * xv[j] / (n * n);
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < CH; ++j) s += u[j] * sg[i * CH + j];
        const float sgn = xv[i] > 0.0f ? 1.0f : (xv[i] < 0.0f ? -1.0f : 0.0f);
        dst[i] = (act_kind == 1 ? g[i] * nrm[i] : g[i] / nrm[i]) + sgn * s;
      }
    } else {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        float m = 1.0f;
        if (act_kind == 3) m = xv[i] > 0.0f ? 1.0f : 0.0f;
        if (act_kind == 4) m = xv[i] >= 0.0f ? 1.0f : 0.2f;
        dst[i] = g[i] * m;
      }
    }
    if (has_res) {
#pragma unroll
      for (int i = 0; i < CH; ++i) dst[CH + i] = g[i];
    }
    for (int i = c2; i < cp; ++i) dst[i] = 0.0f;
  }
}

// ---- input gradient of the 5x5 / stride-2 output layer of the two-layer decoders (Conv2DTranspose Ch -> 3, SAME):
//   g_h[n, i, j, c] = sum_{ky,kx,o} g_x[n, 2i + ky - 1, 2j + kx - 1, o] W[ky, kx, o, c]
// 75 Ch multiply-adds per half-resolution pixel on 3-channel data: an HBM stream, not a GEMM -- one thread per output
// pixel, weights through scalar loads, the 5x5x3 window read through L1 (the gather-GEMM needs 0.48 ms for 5 x 1216^2, this ~0.1 ms).
template <int CH>
__global__ void __launch_bounds__(256) out_adjoint_kernel(const float* __restrict__ gx, int hh, int wh, const float* __restrict__ w2,
                                                          float* __restrict__ gh) {
  const int img = blockIdx.y;
  const int H = 2 * hh, W = 2 * wh;
  const float* src = gx + (size_t)img * H * W * 3;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hh * wh; p += gridDim.x * blockDim.x) {
    const int i = p / wh, j = p - i * wh;
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      const int y = 2 * i + ky - 1;
      if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        const int x = 2 * j + kx - 1;
        if ((unsigned)x >= (unsigned)W) continue;
        const float* g = src + ((size_t)y * W + x) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        const float* wk = w2 + (ky * 5 + kx) * 3 * CH;      // wave-uniform, compile-time offsets: scalar loads
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] += g0 * wk[c] + g1 * wk[CH + c] + g2 * wk[2 * CH + c];
      }
    }
    float* dst = gh + ((size_t)img * hh * wh + p) * CH;
#pragma unroll
    for (int c = 0; c < CH; c += 4) *reinterpret_cast<f32x4*>(dst + c) = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
  }
}

// ---- Keras Adam (non-amsgrad): alpha = lr sqrt(1-b2^t)/(1-b1^t); p -= alpha m / (sqrt(v) + eps) ----
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float alpha, float b1, float b2, float eps,
                                                   float gscale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = gscale * g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= alpha * mi / (sqrtf(vi) + eps);
  }
}


// ---- training-mode entropy terms: the sample is y + U(-.5, .5) (uq method "unoise",
// mshyper/models.py:253-256,277-280 with training=True); rate and partial derivatives at the given sample ----
__device__ __forceinline__ void normal_rate_terms(float v, float raw, float* bits, float* dbits_dv, float* dbits_draw) {
  const float e = expf(raw);
  const float idx = fminf(fmaxf(e, 0.0f), 63.0f);
  const float sigma = expf(kLogScaleMin + kScaleFactor * idx);
  const float hi = (v + 0.5f) / sigma, lo = (v - 0.5f) / sigma;
  const bool right = hi > 0.0f;
  const float big = log_ndtr_f(right ? -lo : hi), small = log_ndtr_f(right ? -hi : lo);
  const float logp = big + log1pf(-expf(small - big));
  const float r_hi = expf(-0.5f * hi * hi - 0.91893853320467274f - logp);
  const float r_lo = expf(-0.5f * lo * lo - 0.91893853320467274f - logp);
  const float dsig_draw = scale_index_gate(e, (r_hi * hi - r_lo * lo) / sigma) ? sigma * kScaleFactor * e : 0.0f;
  *bits = -logp * kInvLn2;
  *dbits_dv = -(r_hi - r_lo) / sigma * kInvLn2;
  *dbits_draw = (r_hi * hi - r_lo * lo) / sigma * dsig_draw * kInvLn2;
}

__global__ void __launch_bounds__(256) noisy_normal_kernel(const float* __restrict__ y_tilde, const float* __restrict__ hyper,
                                                           int64_t hw, int c, float* __restrict__ dbits_dv,
                                                           float* __restrict__ dbits_draw, double* __restrict__ bits) {
  const int img = blockIdx.y;
  const int64_t per = hw * c;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c;
    const int ch = (int)(i - p * c);
    const int64_t gi = img * per + i;
    const float mu = hyper[(img * hw + p) * 2 * c + ch];
    const float raw = hyper[(img * hw + p) * 2 * c + c + ch];
    float b, dv, dr;
    normal_rate_terms(y_tilde[gi] - mu, raw, &b, &dv, &dr);
    dbits_dv[gi] = dv;
    dbits_draw[gi] = dr;
    acc += (double)b;
  }
  block_sum_to(acc, bits + img);
}

// d log p / d L(v + .5), d log p / d L(v - .5) and log p of the noisy deep-factorized density at v
__device__ __forceinline__ void df_logp_terms(float hi, float lo, float* logp, float* dhi, float* dlo) {
  const bool right = hi > 0.0f;
  const float big = log_sigmoid_f(right ? -lo : hi), small = log_sigmoid_f(right ? -hi : lo);
  const float ratio = expf(small - big);
  *logp = big + log1pf(-ratio);
  const float inv = 1.0f / (1.0f - ratio);
  if (!right) {                      // p = s(hi) (1 - ratio), ratio = s(lo) / s(hi)
    *dhi = sigmoid_f(-hi) * inv;
    *dlo = -ratio * sigmoid_f(-lo) * inv;
  } else {                           // p = s(-lo) (1 - ratio), ratio = s(-hi) / s(-lo)
    *dhi = ratio * sigmoid_f(hi) * inv;
    *dlo = -sigmoid_f(lo) * inv;
  }
}

__global__ void __launch_bounds__(256) noisy_factorized_kernel(const float* __restrict__ rec_all, DFDesc d,
                                                               const float* __restrict__ z_tilde, int64_t hw, int c,
                                                               float* __restrict__ dbits_dz, double* __restrict__ bits) {
  const int img = blockIdx.y;
  const int64_t per = hw * c;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const int64_t gi = img * per + i;
    const float v = z_tilde[gi];
    const float* rec = rec_all + (size_t)ch * d.stride;
    float hi, dhi, lo, dlo, logp, ghi, glo;
    df_logits_grad(rec, d, v + 0.5f, &hi, &dhi);
    df_logits_grad(rec, d, v - 0.5f, &lo, &dlo);
    df_logp_terms(hi, lo, &logp, &ghi, &glo);
    dbits_dz[gi] = -(ghi * dhi + glo * dlo) * kInvLn2;
    acc += (double)(-logp * kInvLn2);
  }
  block_sum_to(acc, bits + img);
}

// Reverse pass through the logits network for one input x and upstream gradient gL: accumulates into the
// statically indexed gm / gb / gf (record layout: softplus(matrix), bias, tanh(factor)).
__device__ __forceinline__ void df_logits_param_grad(const float* __restrict__ rec, const DFDesc& d, float x, float gL,
                                                     float (&gm)[kMaxL][kMaxW * kMaxW], float (&gb)[kMaxL][kMaxW],
                                                     float (&gf)[kMaxL][kMaxW]) {
  float hin[kMaxL][kMaxW], th[kMaxL][kMaxW];
  float hv[kMaxW] = {x, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kMaxL; ++k) {
    if (k < d.nl) {
      const int fi = d.w[k], fo = d.w[k + 1];
      float nv[kMaxW] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < kMaxW; ++i) hin[k][i] = hv[i];
#pragma unroll
      for (int o = 0; o < kMaxW; ++o) {
        th[k][o] = 0.f;
        if (o < fo) {
          float s = rec[d.off_b[k] + o];
#pragma unroll
          for (int i = 0; i < kMaxW; ++i)
            if (i < fi) s += rec[d.off_m[k] + o * fi + i] * hv[i];
          if (k < d.nl - 1) {
            th[k][o] = tanhf(s);
            s += rec[d.off_f[k] + o] * th[k][o];
          }
          nv[o] = s;
        }
      }
#pragma unroll
      for (int o = 0; o < kMaxW; ++o) hv[o] = nv[o];
    }
  }
  float gh[kMaxW] = {gL, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = kMaxL - 1; k >= 0; --k) {
    if (k < d.nl) {
      const int fi = d.w[k], fo = d.w[k + 1];
      float gin[kMaxW] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < kMaxW; ++o) {
        if (o < fo) {
          float gs = gh[o];
          if (k < d.nl - 1) {
            gf[k][o] += gs * th[k][o];
            gs *= 1.0f + rec[d.off_f[k] + o] * (1.0f - th[k][o] * th[k][o]);
          }
          gb[k][o] += gs;
#pragma unroll
          for (int i = 0; i < kMaxW; ++i)
            if (i < fi) {
              gm[k][o * kMaxW + i] += gs * hin[k][i];
              gin[i] += gs * rec[d.off_m[k] + o * fi + i];
            }
        }
      }
#pragma unroll
      for (int i = 0; i < kMaxW; ++i) gh[i] = gin[i];
    }
  }
}

// grad_rec[ch][.] += sum over this block's pixels of d bits / d record; thread = (channel, pixel row of 4)
__global__ void __launch_bounds__(256) factorized_param_grad_kernel(const float* __restrict__ rec_all, DFDesc d,
                                                                    const float* __restrict__ z_tilde, int64_t npix, int c,
                                                                    int64_t pslab, float* __restrict__ grad_rec) {
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  if (ch >= c) return;
  const float* rec = rec_all + (size_t)ch * d.stride;
  float gm[kMaxL][kMaxW * kMaxW], gb[kMaxL][kMaxW], gf[kMaxL][kMaxW];
#pragma unroll
  for (int k = 0; k < kMaxL; ++k) {
#pragma unroll
    for (int e = 0; e < kMaxW * kMaxW; ++e) gm[k][e] = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxW; ++e) { gb[k][e] = 0.f; gf[k][e] = 0.f; }
  }
  const int64_t p0 = blockIdx.y * pslab, p1 = p0 + pslab < npix ? p0 + pslab : npix;
  for (int64_t p = p0 + r; p < p1; p += 4) {
    const float v = z_tilde[p * c + ch];
    float hi, dhi, lo, dlo, logp, ghi, glo;
    df_logits_grad(rec, d, v + 0.5f, &hi, &dhi);
    df_logits_grad(rec, d, v - 0.5f, &lo, &dlo);
    df_logp_terms(hi, lo, &logp, &ghi, &glo);
    df_logits_param_grad(rec, d, v + 0.5f, -ghi * kInvLn2, gm, gb, gf);
    df_logits_param_grad(rec, d, v - 0.5f, -glo * kInvLn2, gm, gb, gf);
  }
  float* dst = grad_rec + (size_t)ch * d.stride;
#pragma unroll
  for (int k = 0; k < kMaxL; ++k) {
    if (k < d.nl) {
      const int fi = d.w[k], fo = d.w[k + 1];
#pragma unroll
      for (int o = 0; o < kMaxW; ++o) {
        if (o < fo) {
#pragma unroll
          for (int i = 0; i < kMaxW; ++i)
            if (i < fi) atomicAdd(dst + d.off_m[k] + o * fi + i, gm[k][o * kMaxW + i]);
          atomicAdd(dst + d.off_b[k] + o, gb[k][o]);
          if (k < d.nl - 1) atomicAdd(dst + d.off_f[k] + o, gf[k][o]);
        }
      }
    }
  }
}

// record <- raw variables (device arrays in the sntc_prior_create layout): softplus(matrix), bias, tanh(factor)
__global__ void __launch_bounds__(256) prior_update_kernel(DFDesc d, int c, const float* __restrict__ mats, const float* __restrict__ bias,
                                                           const float* __restrict__ fac, float* __restrict__ rec) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float* r = rec + (size_t)ch * d.stride;
  size_t pm = 0, pb = 0, pf = 0;
  for (int k = 0; k < d.nl; ++k) {
    const int fi = d.w[k], fo = d.w[k + 1];
    for (int e = 0; e < fo * fi; ++e) {
      const float m = mats[pm + (size_t)ch * fo * fi + e];
      r[d.off_m[k] + e] = m > 30.0f ? m : log1pf(expf(m));
    }
    for (int e = 0; e < fo; ++e) r[d.off_b[k] + e] = bias[pb + (size_t)ch * fo + e];
    if (k < d.nl - 1)
      for (int e = 0; e < fo; ++e) r[d.off_f[k] + e] = tanhf(fac[pf + (size_t)ch * fo + e]);
    pm += (size_t)c * fo * fi;
    pb += (size_t)c * fo;
    if (k < d.nl - 1) pf += (size_t)c * fo;
  }
}

// gradients w.r.t. the raw variables from gradients w.r.t. the record: d softplus = sigmoid, d tanh = 1 - tanh^2
__global__ void __launch_bounds__(256) prior_param_grad_kernel(DFDesc d, int c, const float* __restrict__ mats, const float* __restrict__ fac,
                                                               const float* __restrict__ grec, float w, float* __restrict__ gm,
                                                               float* __restrict__ gb, float* __restrict__ gf) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float* g = grec + (size_t)ch * d.stride;
  size_t pm = 0, pb = 0, pf = 0;
  for (int k = 0; k < d.nl; ++k) {
    const int fi = d.w[k], fo = d.w[k + 1];
    for (int e = 0; e < fo * fi; ++e) {
      const size_t at = pm + (size_t)ch * fo * fi + e;
      gm[at] = w * g[d.off_m[k] + e] * sigmoid_f(mats[at]);
    }
    for (int e = 0; e < fo; ++e) gb[pb + (size_t)ch * fo + e] = w * g[d.off_b[k] + e];
    if (k < d.nl - 1)
      for (int e = 0; e < fo; ++e) {
        const size_t at = pf + (size_t)ch * fo + e;
        const float t = tanhf(fac[at]);
        gf[at] = w * g[d.off_f[k] + e] * (1.0f - t * t);
      }
    pm += (size_t)c * fo * fi;
    pb += (size_t)c * fo;
    if (k < d.nl - 1) pf += (size_t)c * fo;
  }
}

}  // namespace sntc

using namespace sntc;

static int grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int sntc_sga_normal_fwd(const float* y_loc, const float* hyper, int n, int64_t hw, int c, float tau,
                                   const float* noise, uint64_t seed, uint64_t step, float* y_tilde, float* sprime,
                                   float* dbits_dv, float* dbits_draw, double* bits, void* stream) {
  if (!y_loc || !hyper || !y_tilde || !sprime || !dbits_dv || !dbits_draw || !bits)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_sga_normal_fwd: null argument");
  if (n < 1 || hw < 1 || c < 1 || !(tau > 0.0f)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_sga_normal_fwd: bad sizes / tau");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(bits, sizeof(double) * n, s)) return zrc;
  hipLaunchKernelGGL(sga_normal_fwd_kernel, dim3(grid_for(hw * c), n), dim3(256), 0, s, y_loc, hyper, hw, c, tau, noise,
                     (unsigned long long)seed, (unsigned long long)step, y_tilde, sprime, dbits_dv, dbits_draw, bits);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_sga_normal_bwd(const float* g_ytilde, const float* sprime, const float* dbits_dv, const float* dbits_draw,
                                   float weight, int64_t npix, int c, float* g_yloc, float* g_hyper, void* stream) {
  if (!g_ytilde || !dbits_dv || !dbits_draw || !g_yloc || !g_hyper)          // sprime == NULL: additive noise, d sample / d loc = 1
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_sga_normal_bwd: null argument");
  hipLaunchKernelGGL(sga_normal_bwd_kernel, dim3(grid_for(npix * c)), dim3(256), 0, (hipStream_t)stream, g_ytilde, sprime,
                     dbits_dv, dbits_draw, weight, npix, c, g_yloc, g_hyper);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_sga_factorized_fwd(const sntc_prior* prior, const float* z_loc, int n, int64_t hw, float tau,
                                       const float* noise, uint64_t seed, uint64_t step, float* z_tilde, float* sprime,
                                       float* dbits_dz, double* bits, void* stream) {
  if (!prior || !z_loc || !z_tilde || !sprime || !dbits_dz || !bits)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_sga_factorized_fwd: null argument");
  if (n < 1 || hw < 1 || !(tau > 0.0f)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_sga_factorized_fwd: bad sizes / tau");
  const int c = prior->channels;
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(bits, sizeof(double) * n, s)) return zrc;
  hipLaunchKernelGGL(sga_factorized_fwd_kernel, dim3(grid_for(hw * c), n), dim3(256), 0, s, prior->rec,
                     prior->d, z_loc, hw, c, tau, noise, (unsigned long long)seed,
                     (unsigned long long)step, z_tilde, sprime, dbits_dz, bits);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_uq_sample(const float* loc, const float* offset, int64_t npix, int c, int offset_stride, int mode,
                              float param, const float* noise, uint64_t seed, uint64_t step, float* out, void* stream) {
  if (!loc || !out) return fail(SNTC_ERR_BAD_SHAPE, "sntc_uq_sample: null argument");
  if (npix < 1 || c < 1 || (offset && offset_stride != 0 && offset_stride < c))      // stride 0: one offset per channel, broadcast over pixels
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_uq_sample: bad sizes");
  if (mode < 0 || mode > 3) return fail(SNTC_ERR_UNSUPPORTED, "sntc_uq_sample: mode must be 0 round, 1 unoise, 2 sga, 3 soft_round");
  if (mode == 2 && !(param > 0.0f)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_uq_sample: sga needs tau > 0");
  hipLaunchKernelGGL(uq_sample_kernel, dim3(grid_for(npix * c)), dim3(256), 0, (hipStream_t)stream, loc, offset, npix, c,
                     offset_stride, mode, param, noise, (unsigned long long)seed, (unsigned long long)step, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_sga_chain(const float* g, const float* dbits, const float* sprime, float weight, int64_t total,
                              float* out, void* stream) {
  if (!g || !dbits || !sprime || !out || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_sga_chain: null argument");
  hipLaunchKernelGGL(sga_chain_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, dbits, sprime, weight,
                     total, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_distortion_grad(const float* x, const float* x_hat, int n, int h, int w, int c, int hs, int ws,
                                    float scale, float* g_xhat, double* sse, void* stream) {
  if (!x || !x_hat || !g_xhat || !sse) return fail(SNTC_ERR_BAD_SHAPE, "sntc_distortion_grad: null argument");
  if (n < 1 || h < 1 || w < 1 || c < 1 || hs < h || ws < w) return fail(SNTC_ERR_BAD_SHAPE, "sntc_distortion_grad: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(sse, sizeof(double) * n, s)) return zrc;
  int b = grid_for((int64_t)hs * ws * c);
  if (b > 512) b = 512;
  hipLaunchKernelGGL(distortion_grad_kernel, dim3(b, n), dim3(256), 0, s, x, x_hat, h, w, c, hs, ws, scale, g_xhat, sse);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

template <int CH>
static int launch_tail_bwd(const float* t, const float* gh, int64_t npix, int has_res, int act_kind, const float* beta,
                           const float* gamma, int cp, float* gt, float* absx, float* gx, hipStream_t s) {
  hipLaunchKernelGGL((tail_bwd_kernel<CH>), dim3(grid_for(npix)), dim3(256), 0, s, t, gh, npix, has_res, act_kind, beta, gamma,
                     cp, gt, absx, gx);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_two_layer_tail_bwd(const float* t, const float* g_h, int64_t npix, int ch, int has_res, int act_kind,
                                       const float* beta, const float* gamma, int cp, float* g_t, float* abs_x, float* g_x,
                                       void* stream) {
  if (!t || !g_h || !g_t || npix < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_bwd: null argument");
  if (cp < ch * (has_res ? 2 : 1)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_bwd: padded channel count too small");
  if ((act_kind == 1 || act_kind == 2) && (!beta || !gamma))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_bwd: GDN parameters missing");
  if ((abs_x == nullptr) != (g_x == nullptr)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_tail_bwd: abs_x and g_x go together");
  hipStream_t s = (hipStream_t)stream;
  switch (ch) {
    case 12: return launch_tail_bwd<12>(t, g_h, npix, has_res, act_kind, beta, gamma, cp, g_t, abs_x, g_x, s);
    case 24: return launch_tail_bwd<24>(t, g_h, npix, has_res, act_kind, beta, gamma, cp, g_t, abs_x, g_x, s);
    case 48: return launch_tail_bwd<48>(t, g_h, npix, has_res, act_kind, beta, gamma, cp, g_t, abs_x, g_x, s);
    default: return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_tail_bwd: hidden channels must be 12, 24 or 48");
  }
}

extern "C" int sntc_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, int64_t t, float grad_scale, void* stream) {
  if (!param || !grad || !m || !v || n < 1 || t < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_adam_step: bad argument");
  const double alpha = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, (float)alpha,
                     beta1, beta2, eps, grad_scale);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

// ---- training-mode entropy entry points ----
extern "C" int sntc_noisy_normal(const float* y_tilde, const float* hyper, int n, int64_t hw, int c, float* dbits_dv,
                                 float* dbits_draw, double* bits, void* stream) {
  if (!y_tilde || !hyper || !dbits_dv || !dbits_draw || !bits) return fail(SNTC_ERR_BAD_SHAPE, "sntc_noisy_normal: null argument");
  if (n < 1 || hw < 1 || c < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_noisy_normal: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(bits, sizeof(double) * n, s)) return zrc;
  hipLaunchKernelGGL(noisy_normal_kernel, dim3(grid_for(hw * c), n), dim3(256), 0, s, y_tilde, hyper, hw, c, dbits_dv, dbits_draw, bits);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_prior_record_floats(const sntc_prior* prior) { return prior ? prior->channels * prior->d.stride : -1; }

extern "C" int sntc_noisy_factorized(const sntc_prior* prior, const float* z_tilde, int n, int64_t hw, float* dbits_dz,
                                     float* grad_record, double* bits, void* stream) {
  if (!prior || !z_tilde || !dbits_dz || !bits) return fail(SNTC_ERR_BAD_SHAPE, "sntc_noisy_factorized: null argument");
  if (n < 1 || hw < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_noisy_factorized: bad sizes");
  const int c = prior->channels;
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(bits, sizeof(double) * n, s)) return zrc;
  hipLaunchKernelGGL(noisy_factorized_kernel, dim3(grid_for(hw * c), n), dim3(256), 0, s, prior->rec, prior->d, z_tilde, hw, c,
                     dbits_dz, bits);
  SNTC_HIP(hipGetLastError());
  if (grad_record) {                         // sum over ALL elements of d bits / d record (caller applies the loss weight)
    if (int zrc = zero_async(grad_record, sizeof(float) * (size_t)c * prior->d.stride, s)) return zrc;
    const int64_t npix = (int64_t)n * hw;
    const int64_t slabs = std::min<int64_t>(256, (npix + 3) / 4);       // few elements, heavy threads: spread them wide
    const int64_t pslab = (npix + slabs - 1) / slabs;
    hipLaunchKernelGGL(factorized_param_grad_kernel, dim3((c + 63) / 64, (unsigned)slabs), dim3(256), 0, s, prior->rec, prior->d,
                       z_tilde, npix, c, pslab, grad_record);
    SNTC_HIP(hipGetLastError());
  }
  return SNTC_OK;
}

extern "C" int sntc_prior_update(sntc_prior* prior, const float* matrices, const float* biases, const float* factors, void* stream) {
  if (!prior || !matrices || !biases || (prior->d.nl > 1 && !factors)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_prior_update: null argument");
  hipLaunchKernelGGL(prior_update_kernel, dim3((prior->channels + 255) / 256), dim3(256), 0, (hipStream_t)stream, prior->d,
                     prior->channels, matrices, biases, factors, prior->rec);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_prior_param_grad(const sntc_prior* prior, const float* matrices, const float* factors, const float* grad_record,
                                     float weight, float* g_matrices, float* g_biases, float* g_factors, void* stream) {
  if (!prior || !matrices || !grad_record || !g_matrices || !g_biases || (prior->d.nl > 1 && (!factors || !g_factors)))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_prior_param_grad: null argument");
  hipLaunchKernelGGL(prior_param_grad_kernel, dim3((prior->channels + 255) / 256), dim3(256), 0, (hipStream_t)stream, prior->d,
                     prior->channels, matrices, factors, grad_record, weight, g_matrices, g_biases, g_factors);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

template <int CH>
static int launch_out_adjoint(const float* gx, int n, int hh, int wh, const float* w2, float* gh, hipStream_t s) {
  const int blocks = std::min((hh * wh + 255) / 256, 4096);
  hipLaunchKernelGGL((out_adjoint_kernel<CH>), dim3(blocks, n), dim3(256), 0, s, gx, hh, wh, w2, gh);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_two_layer_out_adjoint(const float* g_xhat, int n, int hh, int wh, int ch, const float* w2, int k2, int s2,
                                          int cout, float* g_h, void* stream) {
  if (!g_xhat || !w2 || !g_h || n < 1 || hh < 1 || wh < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_out_adjoint: bad argument");
  if (k2 != 5 || s2 != 2 || cout != 3) return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_out_adjoint: 5x5 / stride-2 / 3-channel output layer only");
  if (n > 65535) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_out_adjoint: batch too large");
  hipStream_t s = (hipStream_t)stream;
  switch (ch) {
    case 12: return launch_out_adjoint<12>(g_xhat, n, hh, wh, w2, g_h, s);
    case 24: return launch_out_adjoint<24>(g_xhat, n, hh, wh, w2, g_h, s);
    case 48: return launch_out_adjoint<48>(g_xhat, n, hh, wh, w2, g_h, s);
    default: return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_out_adjoint: hidden channels must be 12, 24 or 48");
  }
}

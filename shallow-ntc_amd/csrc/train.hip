// train.hip -- element-wise pieces of the training step (Model.train_step,
// mshyper/models.py:375-383 with frame_loss_given_latent_rvs(training=True)).  All HBM-bound streams; the
// contractions of the backward pass are sntc_conv_forward (input gradients, adjoint plans) and sntc_conv_wgrad.
#include <algorithm>
#include <cmath>
#include "device_math.h"

namespace sntc {

#define SNTC_GRID_STRIDE(i, total) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// g_pre = g * act'(.) expressed through the layer OUTPUT y: relu 1[y > 0]; leaky_relu(0.2) 1 or 0.2; sigmoid y (1 - y)
__global__ void __launch_bounds__(256) act_backward_kernel(const float* __restrict__ g, const float* __restrict__ y, int64_t total,
                                                           int act, float* __restrict__ out) {
  SNTC_GRID_STRIDE(i, total) {
    const float yv = y[i];
    float d = 1.0f;
    if (act == SNTC_ACT_RELU) d = yv > 0.0f ? 1.0f : 0.0f;
    else if (act == SNTC_ACT_LEAKY_RELU) d = yv > 0.0f ? 1.0f : 0.2f;
    else if (act == SNTC_ACT_SIGMOID) d = yv * (1.0f - yv);
    out[i] = g[i] * d;
  }
}

// SimpleAttention gate (elic.py:97-100), unfused for training: out = x + t * s
__global__ void __launch_bounds__(256) gate_forward_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                           const float* __restrict__ s, int64_t total, float* __restrict__ out) {
  SNTC_GRID_STRIDE(i, total) out[i] = x[i] + t[i] * s[i];
}
// g_t = g s;  g_spre = g t s (1 - s)  (through the sigmoid of the gate conv)
__global__ void __launch_bounds__(256) gate_backward_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                            const float* __restrict__ s, int64_t total, float* __restrict__ g_t,
                                                            float* __restrict__ g_spre) {
  SNTC_GRID_STRIDE(i, total) {
    const float gv = g[i], sv = s[i];
    g_t[i] = gv * sv;
    g_spre[i] = gv * t[i] * sv * (1.0f - sv);
  }
}

__global__ void __launch_bounds__(256) axpy_kernel(float* __restrict__ a, const float* __restrict__ b, float alpha, int64_t total) {
  SNTC_GRID_STRIDE(i, total) a[i] += alpha * b[i];
}

// out = x + u, u ~ U(-.5, .5): counter-based (seed, step, element) unless a noise tensor is given (tests)
__global__ void __launch_bounds__(256) noise_add_kernel(const float* __restrict__ x, int64_t total, const float* __restrict__ noise,
                                                        unsigned long long seed, unsigned long long step, float* __restrict__ out) {
  SNTC_GRID_STRIDE(i, total) {
    float u;
    if (noise) u = noise[i];
    else {
      u = uniform01(splitmix64(stream_key(seed, step) + (unsigned long long)i)) - 0.5f;
    }
    out[i] = x[i] + u;
  }
}

__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, int64_t total, double* __restrict__ out) {
  double acc = 0.0;
  SNTC_GRID_STRIDE(i, total) {
    const float v = x[i];
    acc += (double)v * (double)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// Hidden layer of the two-layer decoders, unfused for training: h = act(t[..., :CH]) (+ t[..., CH:2CH])
// act_kind as in sntc_two_layer_tail (1 = IGDN1: y_j = x_j (beta_j + sum_i |x_i| gamma_ij))
template <int CH>
__global__ void __launch_bounds__(256) hidden_forward_kernel(const float* __restrict__ t, int64_t npix, int has_res, int act_kind,
                                                             const float* __restrict__ beta, const float* __restrict__ gamma,
                                                             float* __restrict__ h) {
  __shared__ float sg[CH * CH];
  __shared__ float sb[CH];
  const bool use_gdn = act_kind == 1 || act_kind == 2;
  if (use_gdn) {
    for (int i = threadIdx.x; i < CH * CH; i += blockDim.x) sg[i] = gamma[i];
    for (int i = threadIdx.x; i < CH; i += blockDim.x) sb[i] = beta[i];
  }
  __syncthreads();
  const int c2 = has_res ? 2 * CH : CH;
  SNTC_GRID_STRIDE(p, npix) {
    float xv[CH];
#pragma unroll
    for (int i = 0; i < CH; i += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(t + p * c2 + i);
      xv[i] = a[0]; xv[i + 1] = a[1]; xv[i + 2] = a[2]; xv[i + 3] = a[3];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float v = xv[j];
      if (use_gdn) {
        float nrm = sb[j];
#pragma unroll
        for (int i = 0; i < CH; ++i) nrm += fabsf(xv[i]) * sg[i * CH + j];
        v = act_kind == 1 ? v * nrm : v / nrm;
      } else if (act_kind == 3) v = fmaxf(v, 0.0f);
      else if (act_kind == 4) v = v > 0.0f ? v : 0.2f * v;
      if (has_res) v += t[p * c2 + CH + j];
      h[p * CH + j] = v;
    }
  }
}

// tfc GDNParameter (non-negative reparameterisation): eff = max(raw, bound)^2 - pedestal;
// backward with the "identity_if_towards" rule of tfc.lower_bound: the gradient passes when raw >= bound or when it
// pushes raw up (g_raw < 0)
__global__ void __launch_bounds__(256) reparam_forward_kernel(const float* __restrict__ raw, int64_t total, float bound, float pedestal,
                                                              float* __restrict__ eff) {
  SNTC_GRID_STRIDE(i, total) {
    const float r = fmaxf(raw[i], bound);
    eff[i] = r * r - pedestal;
  }
}
__global__ void __launch_bounds__(256) reparam_backward_kernel(const float* __restrict__ raw, const float* __restrict__ g_eff,
                                                               int64_t total, float bound, float* __restrict__ g_raw) {
  SNTC_GRID_STRIDE(i, total) {
    const float rv = raw[i];
    const float g = g_eff[i] * 2.0f * fmaxf(rv, bound);
    g_raw[i] = (rv >= bound || g < 0.0f) ? g : 0.0f;
  }
}

// ---- GDN / IGDN layers of the analysis / synthesis stacks (tfc.GDN with alpha = 1, epsilon = 1 -- the reference's GDN1,
// common/transforms.py:8-63): norm_j = beta_j + sum_i |x_i| gamma_ij comes from a 1x1 gather-GEMM plan (|x| prologue);
// the element-wise rest of forward and backward lives here.
//   forward:  y = x / norm                 inverse: y = x * norm
//   backward: q = d loss / d norm = -g x / norm^2   (inverse: g x);   t = q gamma^T (adjoint 1x1 plan);
//             dx = g / norm (inverse: g norm) + sign(x) t;   d beta = sum_p q;   d gamma_ij = sum_p |x_i| q_j
__global__ void __launch_bounds__(256) gdn_apply_kernel(const float* __restrict__ x, const float* __restrict__ norm, int64_t total,
                                                        int inverse, float* __restrict__ y) {
  SNTC_GRID_STRIDE(i, total) y[i] = inverse ? x[i] * norm[i] : x[i] / norm[i];
}
__global__ void __launch_bounds__(256) gdn_bwd_prep_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                           const float* __restrict__ norm, int64_t total, int inverse,
                                                           float* __restrict__ q, float* __restrict__ absx) {
  SNTC_GRID_STRIDE(i, total) {
    const float n = norm[i];
    q[i] = inverse ? g[i] * x[i] : -g[i] * x[i] / (n * n);
    absx[i] = fabsf(x[i]);
  }
}
__global__ void __launch_bounds__(256) gdn_bwd_finish_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                             const float* __restrict__ norm, const float* __restrict__ t,
                                                             int64_t total, int inverse, float* __restrict__ dx) {
  SNTC_GRID_STRIDE(i, total) {
    const float xv = x[i];
    const float sgn = xv > 0.0f ? 1.0f : (xv < 0.0f ? -1.0f : 0.0f);
    dx[i] = (inverse ? g[i] * norm[i] : g[i] / norm[i]) + sgn * t[i];
  }
}

// out[r, c] = sum_k A[r, k] B[k, c] (transpose_a: A[k, r]) for a SMALL A (<= 96 x 96, e.g. the real-DFT basis of a 9x9
// kernel) and a wide B: tfc.RDFTParameter, kernel = M rdft and d rdft = M^T d kernel (tf_checkpoint.irdft_matrix)
__global__ void __launch_bounds__(256) small_matmul_kernel(const float* __restrict__ A, const float* __restrict__ B, int R, int K,
                                                           int64_t C, int transpose_a, float* __restrict__ out) {
  extern __shared__ float sa[];                       // [R][K]
  for (int i = threadIdx.x; i < R * K; i += blockDim.x) {
    const int r = i / K, k = i - r * K;
    sa[i] = transpose_a ? A[(size_t)k * R + r] : A[i];
  }
  __syncthreads();
  SNTC_GRID_STRIDE(c, C) {
    for (int r = 0; r < R; ++r) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) acc = fmaf(sa[r * K + k], B[(size_t)k * C + c], acc);
      out[(size_t)r * C + c] = acc;
    }
  }
}

// dst[t, b, a] = src[t, a, b]  (the weight gradient of an up-sampling SignalConv2D comes out channel-transposed)
__global__ void __launch_bounds__(256) transpose_last2_kernel(const float* __restrict__ src, int taps, int A, int B,
                                                              float* __restrict__ dst) {
  const int64_t total = (int64_t)taps * A * B;
  SNTC_GRID_STRIDE(i, total) {
    const int b = (int)(i % B);
    const int64_t r = i / B;
    const int a = (int)(r % A);
    const int t = (int)(r / A);
    dst[((int64_t)t * B + b) * A + a] = src[i];
  }
}

}  // namespace sntc

using namespace sntc;

static int tr_grid(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int sntc_act_backward(const float* g, const float* y, int64_t total, int act, float* g_pre, void* stream) {
  if (!g || !y || !g_pre || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_act_backward: bad argument");
  if (act != SNTC_ACT_NONE && act != SNTC_ACT_RELU && act != SNTC_ACT_LEAKY_RELU && act != SNTC_ACT_SIGMOID)
    return fail(SNTC_ERR_UNSUPPORTED, "sntc_act_backward: unknown activation");
  hipLaunchKernelGGL(act_backward_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, g, y, total, act, g_pre);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gate_forward(const float* x, const float* t, const float* s, int64_t total, float* out, void* stream) {
  if (!x || !t || !s || !out || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gate_forward: bad argument");
  hipLaunchKernelGGL(gate_forward_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, x, t, s, total, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gate_backward(const float* g, const float* t, const float* s, int64_t total, float* g_t, float* g_spre,
                                  void* stream) {
  if (!g || !t || !s || !g_t || !g_spre || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gate_backward: bad argument");
  hipLaunchKernelGGL(gate_backward_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, g, t, s, total, g_t, g_spre);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_axpy(float* a, const float* b, float alpha, int64_t total, void* stream) {
  if (!a || !b || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_axpy: bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, total);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_noise_add(const float* x, int64_t total, const float* noise, uint64_t seed, uint64_t step, float* out,
                              void* stream) {
  if (!x || !out || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_noise_add: bad argument");
  hipLaunchKernelGGL(noise_add_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, x, total, noise,
                     (unsigned long long)seed, (unsigned long long)step, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_sumsq(const float* x, int64_t total, double* out, void* stream) {
  if (!x || !out || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_sumsq: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(out, sizeof(double), s)) return zrc;
  hipLaunchKernelGGL(sumsq_kernel, dim3(std::min(tr_grid(total), 1024)), dim3(256), 0, s, x, total, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

template <int CH>
static int launch_hidden(const float* t, int64_t npix, int has_res, int act_kind, const float* beta, const float* gamma, float* h,
                         hipStream_t s) {
  hipLaunchKernelGGL((hidden_forward_kernel<CH>), dim3(tr_grid(npix)), dim3(256), 0, s, t, npix, has_res, act_kind, beta, gamma, h);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_two_layer_hidden(const float* t, int64_t npix, int ch, int has_res, int act_kind, const float* beta,
                                     const float* gamma, float* h, void* stream) {
  if (!t || !h || npix < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_hidden: bad argument");
  if ((act_kind == 1 || act_kind == 2) && (!beta || !gamma)) return fail(SNTC_ERR_BAD_SHAPE, "sntc_two_layer_hidden: GDN parameters missing");
  hipStream_t s = (hipStream_t)stream;
  switch (ch) {
    case 12: return launch_hidden<12>(t, npix, has_res, act_kind, beta, gamma, h, s);
    case 24: return launch_hidden<24>(t, npix, has_res, act_kind, beta, gamma, h, s);
    case 48: return launch_hidden<48>(t, npix, has_res, act_kind, beta, gamma, h, s);
    default: return fail(SNTC_ERR_UNSUPPORTED, "sntc_two_layer_hidden: hidden channels must be 12, 24 or 48");
  }
}

extern "C" int sntc_gdn_reparam_forward(const float* raw, int64_t total, float bound, float pedestal, float* eff, void* stream) {
  if (!raw || !eff || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gdn_reparam_forward: bad argument");
  hipLaunchKernelGGL(reparam_forward_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, raw, total, bound, pedestal, eff);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gdn_reparam_backward(const float* raw, const float* g_eff, int64_t total, float bound, float* g_raw, void* stream) {
  if (!raw || !g_eff || !g_raw || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gdn_reparam_backward: bad argument");
  hipLaunchKernelGGL(reparam_backward_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, raw, g_eff, total, bound, g_raw);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gdn_apply(const float* x, const float* norm, int64_t total, int inverse, float* y, void* stream) {
  if (!x || !norm || !y || total < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gdn_apply: bad argument");
  if (total == 0) return SNTC_OK;
  hipLaunchKernelGGL(gdn_apply_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, x, norm, total, inverse, y);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gdn_backward_prep(const float* g, const float* x, const float* norm, int64_t total, int inverse, float* q,
                                      float* abs_x, void* stream) {
  if (!g || !x || !norm || !q || !abs_x || total < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gdn_backward_prep: bad argument");
  if (total == 0) return SNTC_OK;
  hipLaunchKernelGGL(gdn_bwd_prep_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, g, x, norm, total, inverse, q, abs_x);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_gdn_backward_finish(const float* g, const float* x, const float* norm, const float* t, int64_t total,
                                        int inverse, float* dx, void* stream) {
  if (!g || !x || !norm || !t || !dx || total < 0) return fail(SNTC_ERR_BAD_SHAPE, "sntc_gdn_backward_finish: bad argument");
  if (total == 0) return SNTC_OK;
  hipLaunchKernelGGL(gdn_bwd_finish_kernel, dim3(tr_grid(total)), dim3(256), 0, (hipStream_t)stream, g, x, norm, t, total, inverse, dx);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_small_matmul(const float* a, const float* b, int rows, int k, int64_t cols, int transpose_a, float* out,
                                 void* stream) {
  if (!a || !b || !out || rows < 1 || k < 1 || cols < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_small_matmul: bad argument");
  if ((size_t)rows * k * sizeof(float) > 48 * 1024) return fail(SNTC_ERR_UNSUPPORTED, "sntc_small_matmul: the left matrix must fit 48 KB of LDS");
  hipLaunchKernelGGL(small_matmul_kernel, dim3(tr_grid(cols)), dim3(256), (size_t)rows * k * sizeof(float), (hipStream_t)stream, a, b,
                     rows, k, cols, transpose_a, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_transpose_last2(const float* src, int taps, int a, int b, float* dst, void* stream) {
  if (!src || !dst || taps < 1 || a < 1 || b < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_transpose_last2: bad argument");
  hipLaunchKernelGGL(transpose_last2_kernel, dim3(tr_grid((int64_t)taps * a * b)), dim3(256), 0, (hipStream_t)stream, src, taps, a, b, dst);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

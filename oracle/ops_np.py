"""float64 NumPy restatement of the dependency ops on the shallow-ntc hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: TensorFlow /
tensorflow-compression / tensorflow-probability are absent, so each function
restates the *published* semantics of the op the reference calls and cites the
reference call site.  All tensors are NHWC; all math is float64.

Deliberately written in the "obviously correct" scatter/gather-by-tap form, not
the form the HIP kernels use (phase-decomposed gather GEMM).
"""
from __future__ import annotations

import math

import numpy as np
from scipy import special as sp

F64 = np.float64


# --------------------------------------------------------------------------
# activations  (reference common/transforms.py:66-78, common/elic.py:253-270)
# --------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, 0.0)


def leaky_relu(x, alpha=0.2):
    """tf.nn.leaky_relu default alpha=0.2 (transforms.py:76-78 -> getattr(tf.nn, 'leaky_relu'))."""
    return np.where(x >= 0, x, alpha * x)


def sigmoid(x):
    return sp.expit(x)


def channels(x):
    return x.shape[-1]


def as_input(x):
    return np.asarray(x, F64)


def append_ones(x):
    """JPEGLikeSynthesis use_offset (transforms.py:291-293)."""
    return np.concatenate([x, np.ones(x.shape[:3] + (1,), F64)], axis=-1)


def depth_to_space(x, block=2):
    """tf.nn.depth_to_space, NHWC (the upsampling steps of TwoLayerResSynthesis(res_type="d2s"), common/transforms.py:341-348):
    input channel (dy * block + dx) * C + c -> output pixel (iy * block + dy, ix * block + dx), channel c."""
    n, h, w, c = x.shape
    co = c // (block * block)
    y = x.reshape(n, h, w, block, block, co).transpose(0, 1, 3, 2, 4, 5)
    return y.reshape(n, h * block, w * block, co)


ACTIVATIONS = {None: lambda x: x, "none": lambda x: x, "relu": relu,
               "leaky_relu": leaky_relu, "lrelu": leaky_relu, "sigmoid": sigmoid}


# --------------------------------------------------------------------------
# Keras Conv2D(padding="SAME"), NHWC, kernel HWIO, cross-correlation
# reference call sites: common/transforms.py:81-84,186-191,214-218;
# common/elic.py:61-63,92-93,136-140,253-270.  Semantics: SURVEY.md A.1.
# --------------------------------------------------------------------------
def same_pad(in_size: int, k: int, s: int):
    """TF 'SAME': out=ceil(in/s); total=max((out-1)s+k-in,0); extra pixel goes after."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d(x, w, b=None, stride=1, pad=None):
    """y[n,i,j,o] = b[o] + sum x[n, i*s+ky-pt, j*s+kx-pl, c] * w[ky,kx,c,o]; zeros outside.

    pad=None -> TF/Keras SAME.  pad=((pt,pb),(pl,pr)) -> explicit zero padding then VALID
    (used for tfc.SignalConv2D 'same_zeros').
    """
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, ci = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2, (x.shape, w.shape)
    if pad is None:
        ho, pt, pb = same_pad(h, kh, stride)
        wo, pl, pr = same_pad(wd, kw, stride)
    else:
        (pt, pb), (pl, pr) = pad
        ho = (h + pt + pb - kh) // stride + 1
        wo = (wd + pl + pr - kw) // stride + 1
    xp = np.zeros((n, h + pt + pb + stride, wd + pl + pr + stride, ci), F64)
    xp[:, pt:pt + h, pl:pl + wd] = x
    y = np.zeros((n, ho, wo, co), F64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (ho - 1) * stride + 1:stride, kx:kx + (wo - 1) * stride + 1:stride]
            y += patch @ w[ky, kx]
    if b is not None:
        y += np.asarray(b, F64)
    return y


# --------------------------------------------------------------------------
# Keras Conv2DTranspose(padding="SAME"), kernel [kh,kw,Cout,Cin]
# reference call sites: common/transforms.py:85-90,227-231,284-287,307-313,
# 331-338,354-357,371-373.  Semantics: SURVEY.md A.2 (= conv2d_backprop_input of
# the SAME forward conv that maps in*s -> in).
# --------------------------------------------------------------------------
def conv2d_transpose(x, w, b=None, stride=1, pad_before=None, kernel_layout="OI"):
    """Scatter form: out[n, i*s+ky-pt, j*s+kx-pl, o] += x[n,i,j,c] * W[ky,kx][o,c]; out size = in*s.

    pad_before=None -> Keras SAME: pt = max(k-s,0)//2.
    kernel_layout "OI": w[ky,kx,o,c] (Keras Conv2DTranspose);  "IO": w[ky,kx,c,o]
    (tfc.SignalConv2D corr=False, strides_up -- true convolution, no flip in scatter form).
    """
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, ci = x.shape
    kh, kw = w.shape[:2]
    if kernel_layout == "OI":
        co, ci2 = w.shape[2:]
    else:
        ci2, co = w.shape[2:]
    assert ci == ci2, (x.shape, w.shape, kernel_layout)
    s = stride
    if pad_before is None:
        pt = max(kh - s, 0) // 2
        pl = max(kw - s, 0) // 2
    else:
        pt, pl = pad_before
    fh = max((h - 1) * s + kh, h * s + pt)
    fw = max((wd - 1) * s + kw, wd * s + pl)
    full = np.zeros((n, fh, fw, co), F64)
    for ky in range(kh):
        for kx in range(kw):
            wk = w[ky, kx].T if kernel_layout == "OI" else w[ky, kx]     # [ci, co]
            full[:, ky:ky + (h - 1) * s + 1:s, kx:kx + (wd - 1) * s + 1:s] += x @ wk
    y = full[:, pt:pt + h * s, pl:pl + wd * s].copy()
    if b is not None:
        y += np.asarray(b, F64)
    return y


# --------------------------------------------------------------------------
# tfc.SignalConv2D(padding="same_zeros")  -- SURVEY.md A.3
# reference call sites: common/transforms.py:101-112,123-134,152-155,172-175,240-262
# --------------------------------------------------------------------------
def signal_conv_down(x, w, b=None, stride=1):
    """corr=True, strides_down=s: kernel centred, pad (k//2, (k-1)//2), cross-correlation."""
    kh, kw = w.shape[:2]
    pad = ((kh // 2, (kh - 1) // 2), (kw // 2, (kw - 1) // 2))
    return conv2d(x, w, b, stride=stride, pad=pad)


def signal_conv_up(x, w, b=None, stride=1):
    """corr=False, strides_up=s: zero-stuff (length in*s) then true convolution with the
    kernel centred at (k-1)//2:  out[s*i + j - (k-1)//2] += x[i] * w[j]; kernel [kh,kw,Cin,Cout]."""
    kh, kw = w.shape[:2]
    return conv2d_transpose(x, w, b, stride=stride, pad_before=((kh - 1) // 2, (kw - 1) // 2),
                            kernel_layout="IO")


# --------------------------------------------------------------------------
# GDN  (reference common/transforms.py:8-63 GDN1; :150,170 tfc.GDN) -- SURVEY.md A.4
# --------------------------------------------------------------------------
def gdn(x, beta, gamma, inverse=False, alpha=1, epsilon=1.0):
    """norm = (beta + |x|^alpha @ gamma)^epsilon, gamma indexed [in, out]; y = x/norm or x*norm.

    GDN1 (transforms.py:26-63) is alpha=1, epsilon=1.  alpha=2, epsilon=0.5 is the classic
    Balle-2016 form.  beta/gamma are the *effective* (already un-reparameterised) values.
    """
    x = np.asarray(x, F64)
    beta = np.asarray(beta, F64)
    gamma = np.asarray(gamma, F64)
    if alpha == 1:
        pool = np.abs(x)
    elif alpha == 2:
        pool = x * x
    else:
        raise NotImplementedError(alpha)
    norm = pool @ gamma + beta
    if epsilon == 0.5:
        norm = np.sqrt(norm)
    elif epsilon != 1:
        raise NotImplementedError(epsilon)
    return x * norm if inverse else x / norm


# --------------------------------------------------------------------------
# pixel helpers (reference common/image_utils.py:22-71, common/data_lib.py:24-52)
# --------------------------------------------------------------------------
def pad_images(x, div):
    """Reflect-pad bottom/right so H, W are multiples of div (image_utils.py:41-66; tf.pad REFLECT
    mirrors without repeating the edge sample)."""
    n, h, w, c = x.shape
    ph = -(-h // div) * div - h
    pw = -(-w // div) * div - w
    if ph == 0 and pw == 0:
        return x
    return np.pad(x, ((0, 0), (0, ph), (0, pw), (0, 0)), mode="reflect")


def unpad_images(x, shape):
    return x[:, :shape[1], :shape[2], :]


def normalize_image(img):
    return np.asarray(img, F64) / 255.0 - 0.5


def floats_to_pixels(x, training):
    """data_lib.py:48-52 + image_utils.py:22-23: (x+.5)*255; eval: tf.round (half-to-even) then
    saturate_cast uint8, evaluated in float32 like the reference so ties land identically.
    training: no rounding; kept in float64 so finite differences of the loss are clean."""
    if training:
        return (np.asarray(x, F64) + 0.5) * 255.0
    x = (np.asarray(x, np.float32) + np.float32(0.5)) * np.float32(255.0)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def mse_psnr(x, y, max_val=255.0):
    """image_utils.py:26-38: per-image mean squared difference and
    psnr = -10 (ln mse - 2 ln max)/ln 10."""
    x = np.asarray(x, F64)
    y = np.asarray(y, F64)
    mses = ((x - y) ** 2).reshape(x.shape[0], -1).mean(axis=1)
    with np.errstate(divide="ignore"):
        psnrs = -10.0 * (np.log(mses) - 2.0 * np.log(max_val)) / np.log(10.0)
    return mses, psnrs


# --------------------------------------------------------------------------
# entropy models
# --------------------------------------------------------------------------
NUM_SCALES = 64                      # mshyper/models.py:28
SCALE_MIN = 0.11                     # :29
SCALE_MAX = 256.0                    # :30
SCALE_FACTOR = (math.log(SCALE_MAX) - math.log(SCALE_MIN)) / (NUM_SCALES - 1.0)   # :31


def scale_fn(i):
    """SCALE_FN (mshyper/models.py:32)."""
    return np.exp(math.log(SCALE_MIN) + SCALE_FACTOR * np.asarray(i, F64))


def round_half_even(x):
    """tf.round."""
    return np.rint(x)


def _log_diff_exp(big, small):
    """log(exp(big) - exp(small)) = big + log1p(-exp(small-big))  (tfc uniform_noise adapter)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return big + np.log1p(-np.exp(small - big))


def noisy_logprob_from_logcdf_logsf(logcdf_hi, logcdf_lo, logsf_hi, logsf_lo):
    """tfc UniformNoiseAdapter log_prob: log(cdf(v+.5)-cdf(v-.5)); right of the
    median (logsf(v+.5) < logcdf(v+.5)) use the survival-function pair instead."""
    cond = logsf_hi < logcdf_hi
    big = np.where(cond, logsf_lo, logcdf_hi)
    small = np.where(cond, logsf_hi, logcdf_lo)
    return _log_diff_exp(big, small)


def noisy_normal_logprob(v, sigma):
    """log P(v) under Normal(0, sigma) convolved with U(-.5,.5) (tfc.NoisyNormal;
    mshyper/models.py:246-248,278-291)."""
    v = np.asarray(v, F64)
    hi = (v + 0.5) / sigma
    lo = (v - 0.5) / sigma
    return noisy_logprob_from_logcdf_logsf(sp.log_ndtr(hi), sp.log_ndtr(lo),
                                           sp.log_ndtr(-hi), sp.log_ndtr(-lo))


def scale_indexed_normal(y, loc, indexes, coding_rank=3):
    """tfc.LocationScaleIndexedEntropyModel(NoisyNormal, 64, SCALE_FN)(y, indexes, loc=loc,
    training=False)  (mshyper/models.py:246-248,278-279).

    Returns (y_hat, bits[B], symbols) with symbols = round(y-loc) (integer-valued)."""
    y = np.asarray(y, F64)
    loc = np.asarray(loc, F64)
    idx = np.clip(np.asarray(indexes, F64), 0.0, NUM_SCALES - 1.0)
    sigma = scale_fn(idx)
    v = round_half_even(y - loc)
    logp = noisy_normal_logprob(v, sigma)
    axes = tuple(range(-coding_rank, 0))
    bits = logp.sum(axis=axes) / (-math.log(2.0))
    return v + loc, bits, v


def deep_factorized_logits(x, matrices, biases, factors):
    """tfc.DeepFactorized._logits_cumulative.  x[..., C]; matrices[k]: [C, f_{k+1}, f_k]
    (raw, softplus applied here); biases[k]: [C, f_{k+1}]; factors[k]: [C, f_{k+1}] (raw, tanh here)."""
    h = np.asarray(x, F64)[..., None]                       # [..., C, 1]
    nl = len(matrices)
    for k in range(nl):
        m = np.logaddexp(0.0, np.asarray(matrices[k], F64))  # softplus
        h = np.einsum("cij,...cj->...ci", m, h) + np.asarray(biases[k], F64)
        if k < nl - 1:
            h = h + np.tanh(np.asarray(factors[k], F64)) * np.tanh(h)
    return h[..., 0]


def _log_sigmoid(x):
    return -np.logaddexp(0.0, -x)


def deep_factorized_logprob(v, matrices, biases, factors):
    hi = deep_factorized_logits(v + 0.5, matrices, biases, factors)
    lo = deep_factorized_logits(v - 0.5, matrices, biases, factors)
    return noisy_logprob_from_logcdf_logsf(_log_sigmoid(hi), _log_sigmoid(lo),
                                           _log_sigmoid(-hi), _log_sigmoid(-lo))


def batched_deep_factorized(z, matrices, biases, factors, coding_rank=3):
    """tfc.ContinuousBatchedEntropyModel(NoisyDeepFactorized)(z, training=False)
    (mshyper/models.py:249-255; factorized/models.py:101-105); offset 0."""
    z = np.asarray(z, F64)
    v = round_half_even(z)
    logp = deep_factorized_logprob(v, matrices, biases, factors)
    axes = tuple(range(-coding_rank, 0))
    bits = logp.sum(axis=axes) / (-math.log(2.0))
    return v, bits


# --------------------------------------------------------------------------
# SGA  (reference common/latent_rvs_utils.py:8-48,90-103) -- SURVEY.md A.7
# --------------------------------------------------------------------------
def sga_tau(t, r, ub, lb=1e-8, t0=200.0):
    return float(min(max(ub * math.exp(-r * (float(t) - t0)), lb), ub))


def sga_round(mu, tau, gumbel, offset=None, epsilon=1e-5):
    """Deterministic given the Gumbel noise g[..., 2]:  w = softmax((logits + g)/tau),
    logits = (-atanh(clip(mu-floor))/tau, -atanh(clip(ceil-mu))/tau); out = w0 floor + w1 ceil."""
    mu = np.asarray(mu, F64)
    if offset is not None:
        return sga_round(mu - offset, tau, gumbel, None, epsilon) + offset
    fl = np.floor(mu)
    ce = np.ceil(mu)
    l0 = -np.arctanh(np.clip(mu - fl, -1 + epsilon, 1 - epsilon)) / tau
    l1 = -np.arctanh(np.clip(ce - mu, -1 + epsilon, 1 - epsilon)) / tau
    a0 = (l0 + gumbel[..., 0]) / tau
    a1 = (l1 + gumbel[..., 1]) / tau
    m = np.maximum(a0, a1)
    e0 = np.exp(a0 - m)
    e1 = np.exp(a1 - m)
    return (e0 * fl + e1 * ce) / (e0 + e1)


# --------------------------------------------------------------------------
# SSIM / MS-SSIM  (reference mshyper/models.py:321-336 -> tf.image.ssim / tf.image.ssim_multiscale,
# TF 2.10 image_ops_impl.py).  Restated from the published algorithm: 11x11 Gaussian
# (sigma 1.5, normalised by a softmax over the 2-D grid == outer product of normalised 1-D windows),
# VALID depthwise filtering, k1 = .01, k2 = .03, compensation 1.0, 5 scales with weights
# (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), 2x2 average pooling between scales after SYMMETRIC padding of
# odd sizes at the bottom/right, relu on every factor, mean over channels last.
# --------------------------------------------------------------------------
MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_window(size=11, sigma=1.5):
    c = np.arange(size, dtype=F64) - (size - 1) / 2.0
    g = np.exp(-0.5 * c * c / (sigma * sigma))
    return g / g.sum()


def _filter_valid(x, win):
    """Depthwise separable VALID filter over H and W of [n,h,w,c]."""
    k = len(win)
    n, h, w, c = x.shape
    t = np.zeros((n, h - k + 1, w, c), F64)
    for i in range(k):
        t += win[i] * x[:, i:i + h - k + 1]
    o = np.zeros((n, h - k + 1, w - k + 1, c), F64)
    for j in range(k):
        o += win[j] * t[:, :, j:j + w - k + 1]
    return o


def ssim_per_channel(a, b, max_val=255.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """-> (ssim[n,c], cs[n,c]) spatial means of luminance*cs and of cs."""
    a = np.asarray(a, F64)
    b = np.asarray(b, F64)
    win = _gauss_window(filter_size, filter_sigma)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    m0, m1 = _filter_valid(a, win), _filter_valid(b, win)
    num0 = m0 * m1 * 2.0
    den0 = m0 * m0 + m1 * m1
    lum = (num0 + c1) / (den0 + c1)
    num1 = _filter_valid(a * b, win) * 2.0
    den1 = _filter_valid(a * a + b * b, win)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(axis=(1, 2)), cs.mean(axis=(1, 2))


def ssim(a, b, max_val=255.0):
    return ssim_per_channel(a, b, max_val)[0].mean(axis=-1)


def _avg_pool2_symmetric(x):
    n, h, w, c = x.shape
    if h % 2 or w % 2:
        x = np.pad(x, ((0, 0), (0, h % 2), (0, w % 2), (0, 0)), mode="symmetric")
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))


def ms_ssim(a, b, max_val=255.0, power_factors=MSSSIM_WEIGHTS):
    a = np.asarray(a, F64)
    b = np.asarray(b, F64)
    mcs = []
    for k in range(len(power_factors)):
        if k > 0:
            a, b = _avg_pool2_symmetric(a), _avg_pool2_symmetric(b)
        s, cs = ssim_per_channel(a, b, max_val)
        mcs.append(np.maximum(cs, 0.0))
    mcs.pop()
    stack = np.stack(mcs + [np.maximum(s, 0.0)], axis=-1)            # [n, c, scales]
    return np.prod(stack ** np.asarray(power_factors), axis=-1).mean(axis=-1)


def image_quality(a, b, max_val=255.0):
    """mshyper/models.py:321-331: single-scale SSIM when both sides are < 160, MS-SSIM otherwise;
    msssim_db = -10 log10(1 - msssim)."""
    h, w = a.shape[1], a.shape[2]
    v = ssim(a, b, max_val) if (h < 160 and w < 160) else ms_ssim(a, b, max_val)
    with np.errstate(divide="ignore"):
        return v, -10.0 * np.log10(1.0 - v)

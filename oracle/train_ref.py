"""float64 PyTorch-autograd restatement of the reference training loss (Model.train_step's forward:
mshyper/models.py:234-359 with training=True and the 'unoise' uq method, :375-383) and its gradients.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

It is a *backend module* for oracle.transforms_np (``transform(params, x, be=train_ref)``) whose parameters are
float64 torch tensors with ``requires_grad``; activations are NCHW float64.  ``loss_and_grads`` returns the loss
terms and d loss / d variable for every entry of ``params`` -- what ``tape.gradient(loss, trainable_variables)``
gives the reference -- under caller-supplied uniform noise, so the HIP training step can be compared tensor by
tensor.  Also the Keras-Adam / clipnorm / schedule arithmetic used to check one full parameter update.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import transforms_np as T

LOG_SCALE_MIN = math.log(0.11)
SCALE_FACTOR = (math.log(256.0) - math.log(0.11)) / 63.0
GDN_OFFSET = 2.0 ** -18                      # tfc.GDNParameter reparam offset; pedestal = offset^2
GDN_BETA_MIN = 1e-6


# ---- backend interface of transforms_np --------------------------------------------------------
def channels(x):
    return x.shape[1]


def as_input(x):
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).permute(0, 3, 1, 2).contiguous()


def to_nhwc(x):
    return x.detach().permute(0, 2, 3, 1).contiguous().numpy()


def as_params(params):
    """{name: array} -> {name: float64 tensor} (no gradient): lets the eval forward of model_np run its transforms on
    this backend (library convolutions in float64) at sizes where the NumPy tap loops would take minutes."""
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)) for k, v in params.items()}


def append_ones(x):
    return torch.cat([x, torch.ones_like(x[:, :1])], dim=1)


def depth_to_space(x, block=2):
    """tf.nn.depth_to_space on this backend's NCHW tensors (channel = (dy * block + dx) * C + c, TensorFlow's DCR order)."""
    n, c, h, w = x.shape
    co = c // (block * block)
    y = x.reshape(n, block, block, co, h, w).permute(0, 3, 4, 1, 5, 2)
    return y.reshape(n, co, h * block, w * block)


ACTIVATIONS = {None: lambda x: x, "none": lambda x: x, "relu": F.relu,
               "leaky_relu": lambda x: F.leaky_relu(x, 0.2), "lrelu": lambda x: F.leaky_relu(x, 0.2),
               "sigmoid": torch.sigmoid}


def _same(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d(x, w, b=None, stride=1, pad=None):
    kh, kw = w.shape[:2]
    if pad is None:
        pt, pb = _same(x.shape[2], kh, stride)
        pl, pr = _same(x.shape[3], kw, stride)
    else:
        (pt, pb), (pl, pr) = pad
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w.permute(3, 2, 0, 1), b, stride=stride)


def conv2d_transpose(x, w, b=None, stride=1, pad_before=None, kernel_layout="OI"):
    kh, kw = w.shape[:2]
    s = stride
    pt, pl = (max(kh - s, 0) // 2, max(kw - s, 0) // 2) if pad_before is None else pad_before
    wt = w.permute(3, 2, 0, 1) if kernel_layout == "OI" else w.permute(2, 3, 0, 1)
    full = F.conv_transpose2d(x, wt, None, stride=s)
    h, wd = x.shape[2] * s, x.shape[3] * s
    need_h, need_w = pt + h - full.shape[2], pl + wd - full.shape[3]
    if need_h > 0 or need_w > 0:
        full = F.pad(full, (0, max(need_w, 0), 0, max(need_h, 0)))
    y = full[:, :, pt:pt + h, pl:pl + wd]
    return y if b is None else y + b.view(1, -1, 1, 1)


def signal_conv_down(x, w, b=None, stride=1):
    kh, kw = w.shape[:2]
    return conv2d(x, w, b, stride, pad=((kh // 2, (kh - 1) // 2), (kw // 2, (kw - 1) // 2)))


def signal_conv_up(x, w, b=None, stride=1):
    kh, kw = w.shape[:2]
    return conv2d_transpose(x, w, b, stride, pad_before=((kh - 1) // 2, (kw - 1) // 2), kernel_layout="IO")


def gdn(x, beta, gamma, inverse=False, alpha=1, epsilon=1.0):
    pool = x.abs() if alpha == 1 else x * x
    norm = F.conv2d(pool, gamma.t().reshape(gamma.shape[1], gamma.shape[0], 1, 1), beta)
    if epsilon == 0.5:
        norm = norm.sqrt()
    return x * norm if inverse else x / norm


# ---- entropy terms -------------------------------------------------------------------------------
def deep_factorized_logits(x, mats, biases, factors):
    """x [..., C] -> logits of the cumulative (tfc.DeepFactorized._logits_cumulative)."""
    shape = x.shape
    h = x.reshape(-1, shape[-1]).t().unsqueeze(1)                  # [C, 1, N]
    for k, (m, b) in enumerate(zip(mats, biases)):
        h = torch.matmul(F.softplus(m), h) + b.unsqueeze(-1)       # [C, fo, N]
        if k < len(factors):
            h = h + torch.tanh(factors[k]).unsqueeze(-1) * torch.tanh(h)
    return h.squeeze(1).t().reshape(shape)


def noisy_deep_factorized_bits(v, mats, biases, factors):
    """-log2 [sigmoid(L(v + .5)) - sigmoid(L(v - .5))] element-wise (UniformNoiseAdapter.log_prob, stable form)."""
    hi = deep_factorized_logits(v + 0.5, mats, biases, factors)
    lo = deep_factorized_logits(v - 0.5, mats, biases, factors)
    right = hi > 0
    big = F.logsigmoid(torch.where(right, -lo, hi))
    small = F.logsigmoid(torch.where(right, -hi, lo))
    return -(big + torch.log1p(-torch.exp(small - big))) / math.log(2.0)


class _BoundTowards(torch.autograd.Function):
    """tfc.ops.math_ops.upper_bound / lower_bound with gradient="identity_if_towards" (their default, and what
    LocationScaleIndexedEntropyModel._normalize_indexes uses [DEP]): forward clamps; backward passes the gradient where
    the input is inside the bound OR a descent step would move it towards the bound."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.save_for_backward(x)
        ctx.lo, ctx.hi = lo, hi
        return torch.clamp(x, lo, hi)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        ok = ((x <= ctx.hi) | (g > 0)) & ((x >= ctx.lo) | (g < 0))
        return g * ok.to(g.dtype), None, None


def noisy_normal_bits(v, raw):
    """-log2 [Phi((v + .5)/s) - Phi((v - .5)/s)], s = SCALE_FN(bound(exp(raw), 0, 63)) (mshyper/models.py:28-32,275-279)."""
    sigma = torch.exp(LOG_SCALE_MIN + SCALE_FACTOR * _BoundTowards.apply(torch.exp(raw), 0.0, 63.0))
    hi, lo = (v + 0.5) / sigma, (v - 0.5) / sigma
    right = hi > 0
    big = torch.special.log_ndtr(torch.where(right, -lo, hi))
    small = torch.special.log_ndtr(torch.where(right, -hi, lo))
    return -(big + torch.log1p(-torch.exp(small - big))) / math.log(2.0)


def gdn_effective(raw, minimum):
    """tfc.GDNParameter: max(raw, sqrt(minimum + pedestal))^2 - pedestal (plain clamp: tests keep raw above the bound)."""
    pedestal = GDN_OFFSET ** 2
    return torch.clamp(raw, min=math.sqrt(minimum + pedestal)) ** 2 - pedestal


def gdn_raw(effective, minimum):
    pedestal = GDN_OFFSET ** 2
    return np.sqrt(np.maximum(np.asarray(effective, np.float64) + pedestal, pedestal))


# ---- the training loss -----------------------------------------------------------------------------
def loss_and_grads(transform_config, params, x, noise_z, noise_y, rd_lambda, num_filters=(3, 3), gdn_raw_names=(), uq="unoise",
                   factorized=False):
    """params: {name: ndarray} (Model.get_weights() naming; entries listed in ``gdn_raw_names`` hold the RAW
    reparameterised GDN variable as ``(array, minimum)``).  x NHWC in [-0.5, 0.5]; noise_* NHWC in (-.5, .5).
    -> dict(loss, bpp, mse, bits_z[n], bits_y[n], recon NHWC, grads {name: ndarray})."""
    a = dict(transform_config["analysis"])
    analysis = T.build(a.pop("cls"), **a)
    b = analysis.graph.shapes(3)[1] if hasattr(analysis, "graph") else None
    leaves = OrderedDict()
    eff = {}
    for k, v in params.items():
        if k in gdn_raw_names:
            arr, minimum = v
            t = torch.tensor(np.asarray(arr, np.float64), requires_grad=True)
            leaves[k] = t
            eff[k] = gdn_effective(t, minimum)
        else:
            t = torch.tensor(np.asarray(v, np.float64), requires_grad=True)
            leaves[k] = t
            eff[k] = t
    xt = as_input(x)
    n, _, h, w = xt.shape
    y = analysis(T.sub_params(eff, "analysis/"), xt, be=_SELF)
    b = y.shape[1]
    if factorized:        # factorized/models.py:89-183 with training=True: the deep-factorized prior codes y + noise
        s = dict(transform_config["synthesis"])
        synthesis = T.build(s.pop("cls"), cin=b, **s)
        nl = len(num_filters) + 1
        mats = [eff[f"prior/matrix_{k}"] for k in range(nl)]
        biases = [eff[f"prior/bias_{k}"] for k in range(nl)]
        factors = [eff[f"prior/factor_{k}"] for k in range(nl - 1)]
        y_t = y + as_input(noise_y)
        bits_y = noisy_deep_factorized_bits(y_t.permute(0, 2, 3, 1), mats, biases, factors).sum(dim=(1, 2, 3))
        recon = synthesis(T.sub_params(eff, "synthesis/"), y_t, be=_SELF)
        bpp = bits_y.mean() / (h * w)
        mse = ((255.0 * (xt - recon)) ** 2).mean(dim=(1, 2, 3)).mean()
        loss = bpp + rd_lambda * mse
        grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        return dict(loss=float(loss.detach()), bpp=float(bpp.detach()), mse=float(mse.detach()), bits_z=np.zeros(n),
                    bits_y=bits_y.detach().numpy(), recon=to_nhwc(recon), y=to_nhwc(y), z=None,
                    grads={k: (np.zeros(tuple(leaves[k].shape)) if g is None else g.numpy()) for k, g in zip(leaves, grads)})
    ha = dict(transform_config.get("hyper_analysis", dict(cls="HyperAnalysis", bottleneck_size=b)))
    hs = dict(transform_config.get("hyper_synthesis", dict(cls="HyperSynthesis", bottleneck_size=b)))
    s = dict(transform_config["synthesis"])
    hyper_analysis = T.build(ha.pop("cls"), cin=b, **ha)
    z = hyper_analysis(T.sub_params(eff, "hyper_analysis/"), y, be=_SELF)
    hyper_synthesis = T.build(hs.pop("cls"), cin=z.shape[1], **hs)
    synthesis = T.build(s.pop("cls"), cin=b, **s)
    nl = len(num_filters) + 1
    mats = [eff[f"prior/matrix_{k}"] for k in range(nl)]
    biases = [eff[f"prior/bias_{k}"] for k in range(nl)]
    factors = [eff[f"prior/factor_{k}"] for k in range(nl - 1)]
    z_t = z + as_input(noise_z)
    bits_z = noisy_deep_factorized_bits(z_t.permute(0, 2, 3, 1), mats, biases, factors).sum(dim=(1, 2, 3))
    # 'mixedq' (mshyper/models.py:257-259,281-283): the decoder side sees tfc's straight-through rounding
    z_dec = z + (torch.round(z) - z).detach() if uq == "mixedq" else z_t
    hyper = hyper_synthesis(T.sub_params(eff, "hyper_synthesis/"), z_dec, be=_SELF)
    mu, raw = hyper[:, :b], hyper[:, b:]
    y_t = y + as_input(noise_y)
    bits_y = noisy_normal_bits(y_t - mu, raw).sum(dim=(1, 2, 3))
    y_dec = (y - mu) + (torch.round(y - mu) - (y - mu)).detach() + mu if uq == "mixedq" else y_t
    recon = synthesis(T.sub_params(eff, "synthesis/"), y_dec, be=_SELF)
    bpp = bits_z.mean() / (h * w) + bits_y.mean() / (h * w)
    mse = ((255.0 * (xt - recon)) ** 2).mean(dim=(1, 2, 3)).mean()
    loss = bpp + rd_lambda * mse
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    return dict(loss=float(loss.detach()), bpp=float(bpp.detach()), mse=float(mse.detach()), bits_z=bits_z.detach().numpy(), bits_y=bits_y.detach().numpy(),
                recon=to_nhwc(recon), y=to_nhwc(y), z=to_nhwc(z),
                grads={k: (np.zeros(tuple(leaves[k].shape)) if g is None else g.numpy()) for k, g in zip(leaves, grads)})



# ---- the SGA objective (itinf_train_step's loss, mshyper/models.py:260-268,285-291,397-404) -------------------------------
def sga_round(mu, tau, gumbel, offset=None, epsilon=1e-5):
    """common/latent_rvs_utils.py:8-48 in float64 autograd form; ``gumbel`` [..., 2] is the (fixed) Gumbel noise of the
    RelaxedOneHotCategorical sample: w = softmax((logits + g) / tau), out = w0 floor(mu) + w1 ceil(mu)."""
    if offset is not None:
        return sga_round(mu - offset, tau, gumbel, None, epsilon) + offset
    fl, ce = torch.floor(mu).detach(), torch.ceil(mu).detach()
    l0 = -torch.atanh(torch.clamp(mu - fl, -1 + epsilon, 1 - epsilon)) / tau
    l1 = -torch.atanh(torch.clamp(ce - mu, -1 + epsilon, 1 - epsilon)) / tau
    w = torch.softmax(torch.stack([(l0 + gumbel[..., 0]) / tau, (l1 + gumbel[..., 1]) / tau], dim=-1), dim=-1)
    return w[..., 0] * fl + w[..., 1] * ce


def sga_loss_and_grads(transform_config, params, x, z_loc, y_loc, tau, gumbel_z, gumbel_y, rd_lambda, num_filters=(3, 3)):
    """loss = bpp + lambda * MSE(0-255 floats, unrounded) of frame_loss_given_latent_rvs(training=True) under the 'sga' method
    with the Gumbel noise held fixed, and d loss / d (z_loc, y_loc) by float64 autograd -- the two tensors
    itinf_train_step differentiates with respect to (mshyper/models.py:397-399).  Latents NHWC, noise NHWC + [2]."""
    eff = {k: torch.tensor(np.asarray(v, np.float64)) for k, v in params.items()}
    z = torch.tensor(np.asarray(z_loc, np.float64), requires_grad=True)
    y = torch.tensor(np.asarray(y_loc, np.float64), requires_grad=True)
    gz, gy = torch.tensor(np.asarray(gumbel_z, np.float64)), torch.tensor(np.asarray(gumbel_y, np.float64))
    xt = as_input(x)
    n, _, h, w = xt.shape
    b = y.shape[-1]
    hs = dict(transform_config.get("hyper_synthesis", dict(cls="HyperSynthesis", bottleneck_size=b)))
    sy = dict(transform_config["synthesis"])
    hyper_synthesis = T.build(hs.pop("cls"), cin=z.shape[-1], **hs)
    synthesis = T.build(sy.pop("cls"), cin=b, **sy)
    nl = len(num_filters) + 1
    mats = [eff[f"prior/matrix_{k}"] for k in range(nl)]
    biases = [eff[f"prior/bias_{k}"] for k in range(nl)]
    factors = [eff[f"prior/factor_{k}"] for k in range(nl - 1)]
    z_t = sga_round(z, tau, gz)                                                        # offset 0
    bits_z = noisy_deep_factorized_bits(z_t, mats, biases, factors).sum(dim=(1, 2, 3))
    hyper = hyper_synthesis(T.sub_params(eff, "hyper_synthesis/"), z_t.permute(0, 3, 1, 2), be=_SELF)
    mu, raw = hyper[:, :b].permute(0, 2, 3, 1), hyper[:, b:].permute(0, 2, 3, 1)
    y_t = sga_round(y, tau, gy, offset=mu)
    bits_y = noisy_normal_bits(y_t - mu, raw).sum(dim=(1, 2, 3))
    recon = synthesis(T.sub_params(eff, "synthesis/"), y_t.permute(0, 3, 1, 2), be=_SELF)[:, :, :h, :w]
    bpp = bits_z.mean() / (h * w) + bits_y.mean() / (h * w)
    mse = ((255.0 * (xt - recon)) ** 2).mean(dim=(1, 2, 3)).mean()
    loss = bpp + rd_lambda * mse
    g_z, g_y = torch.autograd.grad(loss, [z, y])
    return dict(loss=float(loss.detach()), bpp=float(bpp.detach()), mse=float(mse.detach()), g_z=g_z.numpy(), g_y=g_y.numpy(),
                bits_z=bits_z.detach().numpy(), bits_y=bits_y.detach().numpy())

# ---- optimizer arithmetic (tf.keras.optimizers.Adam + global_clipnorm, common/schedule.py:155-176) --------
def adam_update(param, grad, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-7):
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    alpha = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return param - alpha * m / (np.sqrt(v) + eps), m, v


def clip_scale(grads, clipnorm):
    norm = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return (1.0 if clipnorm is None or norm <= clipnorm else clipnorm / norm), norm


import sys as _sys  # noqa: E402

_SELF = _sys.modules[__name__]

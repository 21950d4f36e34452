"""Independent float32 PyTorch-CPU restatement of the hot-path ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Two jobs:
  1. second, independent implementation of the SAME/same_zeros pad arithmetic and kernel
     layouts (library ``F.conv2d`` / ``F.conv_transpose2d`` + explicit pads/crops instead of the
     tap loops of ops_np) -- tests require the two to agree;
  2. the timed CPU baseline (``cpu_baseline.kind == "port"``): the reference's arithmetic runs in
     TensorFlow-CPU (Eigen/oneDNN, multi-threaded) which is not installable here, so the same
     graph on PyTorch-CPU (oneDNN, ``torch.get_num_threads()`` threads) stands in for it.

It is a *backend module* for oracle.transforms_np: ``transform(params, x, be=torch_ref)``.
Activations live as NCHW float32 tensors in channels_last memory format; ``as_input`` /
``to_nhwc`` convert at the ends.  Converted weights are cached per source array.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

_wcache: dict = {}


def _cached(arr, tag, fn):
    key = (id(arr), tag)
    hit = _wcache.get(key)
    if hit is None or hit[0] is not arr:
        hit = (arr, fn(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))))
        _wcache[key] = hit
    return hit[1]


def clear_cache():
    _wcache.clear()


def channels(x):
    return x.shape[1]


def as_input(x):
    if isinstance(x, torch.Tensor):
        return x
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)


def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().numpy()


def append_ones(x):
    return torch.cat([x, torch.ones_like(x[:, :1])], dim=1)


def depth_to_space(x, block=2):
    """tf.nn.depth_to_space on NCHW tensors (channel = (dy * block + dx) * C + c, TensorFlow's DCR order)."""
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
    wt = _cached(w, "hwio", lambda t: t.permute(3, 2, 0, 1).contiguous())       # OIHW
    bt = None if b is None else _cached(b, "b", lambda t: t)
    if (pt, pl) == (pb, pr):
        return F.conv2d(x, wt, bt, stride=stride, padding=(pt, pl))
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), wt, bt, stride=stride)


def conv2d_transpose(x, w, b=None, stride=1, pad_before=None, kernel_layout="OI"):
    kh, kw = w.shape[:2]
    s = stride
    if pad_before is None:
        pt, pl = max(kh - s, 0) // 2, max(kw - s, 0) // 2
    else:
        pt, pl = pad_before
    if kernel_layout == "OI":     # Keras [kh,kw,Cout,Cin] -> torch conv_transpose weight [Cin,Cout,kh,kw]
        wt = _cached(w, "T_OI", lambda t: t.permute(3, 2, 0, 1).contiguous())
    else:                         # SignalConv2D [kh,kw,Cin,Cout]
        wt = _cached(w, "T_IO", lambda t: t.permute(2, 3, 0, 1).contiguous())
    full = F.conv_transpose2d(x, wt, None, stride=s)          # size (in-1)s+k, full[i*s+ky] += x[i] w[ky]
    h, wd = x.shape[2] * s, x.shape[3] * s
    need_h, need_w = pt + h - full.shape[2], pl + wd - full.shape[3]
    if need_h > 0 or need_w > 0:
        full = F.pad(full, (0, max(need_w, 0), 0, max(need_h, 0)))
    y = full[:, :, pt:pt + h, pl:pl + wd]
    if b is not None:
        y = y + _cached(b, "b", lambda t: t).view(1, -1, 1, 1)
    return y


def signal_conv_down(x, w, b=None, stride=1):
    kh, kw = w.shape[:2]
    return conv2d(x, w, b, stride, pad=((kh // 2, (kh - 1) // 2), (kw // 2, (kw - 1) // 2)))


def signal_conv_up(x, w, b=None, stride=1):
    kh, kw = w.shape[:2]
    return conv2d_transpose(x, w, b, stride, pad_before=((kh - 1) // 2, (kw - 1) // 2), kernel_layout="IO")


def gdn(x, beta, gamma, inverse=False, alpha=1, epsilon=1.0):
    g = _cached(gamma, "gamma", lambda t: t.t().contiguous().view(t.shape[1], t.shape[0], 1, 1))
    bt = _cached(beta, "b", lambda t: t)
    pool = x.abs() if alpha == 1 else x * x
    norm = F.conv2d(pool, g, bt)
    if epsilon == 0.5:
        norm = norm.sqrt()
    return x * norm if inverse else x / norm


# ------------------------------------------------------------------------------------------
# decode / encode legs used by bench.py's cpu_baseline (float32, NHWC numpy in and out)
# ------------------------------------------------------------------------------------------
LOG_SCALE_MIN = math.log(0.11)
SCALE_FACTOR = (math.log(256.0) - math.log(0.11)) / 63.0


def decode(model_np, params, z_hat, symbols_y, unpadded_hw):
    """(z_hat, q) -> hyper-synthesis -> y_hat = q + mu -> synthesis -> u8 pixels
    (mshyper/models.py:273-298 + data_lib.py:48-52)."""
    from . import transforms_np as T
    with torch.no_grad():
        h = model_np.hyper_synthesis(T.sub_params(params, "hyper_synthesis/"), z_hat, be=_SELF)
        c = h.shape[1] // 2
        y_hat = as_input(symbols_y) + h[:, :c]
        rec = model_np.synthesis(T.sub_params(params, "synthesis/"), y_hat, be=_SELF)
        rec = rec[:, :, :unpadded_hw[0], :unpadded_hw[1]]
        px = torch.clamp(torch.round((rec + 0.5) * 255.0), 0, 255).to(torch.uint8)
    return px.permute(0, 2, 3, 1).contiguous().numpy()


def analysis_only(model_np, params, x):
    from . import transforms_np as T
    with torch.no_grad():
        y = model_np.analysis(T.sub_params(params, "analysis/"), x, be=_SELF)
    return y


import sys as _sys
_SELF = _sys.modules[__name__]


def ms_ssim_torch(a, b, max_val=255.0):
    """Independent float32 restatement of tf.image.ssim_multiscale: 2-D 11x11 kernel from a softmax over
    the grid (as TF's _fspecial_gauss), depthwise F.conv2d VALID, F.avg_pool2d after replicate padding."""
    a = torch.as_tensor(np.asarray(a, np.float32)).permute(0, 3, 1, 2)
    b = torch.as_tensor(np.asarray(b, np.float32)).permute(0, 3, 1, 2)
    c = a.shape[1]
    coords = torch.arange(11, dtype=torch.float32) - 5.0
    g = -0.5 * coords ** 2 / 1.5 ** 2
    k2d = torch.softmax((g.view(1, -1) + g.view(-1, 1)).reshape(-1), 0).view(1, 1, 11, 11).repeat(c, 1, 1, 1)
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    weights = torch.tensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])

    def per_channel(x, y):
        red = lambda t: F.conv2d(t, k2d, groups=c)
        m0, m1 = red(x), red(y)
        num0, den0 = 2 * m0 * m1, m0 * m0 + m1 * m1
        lum = (num0 + c1) / (den0 + c1)
        cs = (2 * red(x * y) - num0 + c2) / (red(x * x + y * y) - den0 + c2)
        return (lum * cs).mean(dim=(2, 3)), cs.mean(dim=(2, 3))

    mcs = []
    for k in range(5):
        if k > 0:
            ph, pw = a.shape[2] % 2, a.shape[3] % 2
            if ph or pw:
                a, b = F.pad(a, (0, pw, 0, ph), mode="replicate"), F.pad(b, (0, pw, 0, ph), mode="replicate")
            a, b = F.avg_pool2d(a, 2), F.avg_pool2d(b, 2)
        s, cs = per_channel(a, b)
        mcs.append(torch.relu(cs))
    mcs[-1] = torch.relu(s)
    return torch.prod(torch.stack(mcs, -1) ** weights, -1).mean(-1).numpy()

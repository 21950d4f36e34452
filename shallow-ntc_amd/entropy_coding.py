"""Bitstream for the mean-scale hyperprior codec: integer CDF tables + rANS on the GPU.

The reference never produces a bitstream (``compression=False`` everywhere, mshyper/models.py:246-251): its
bpp is -sum log2 p.  This module makes ``decode`` a real codec: ``Model.compress(x) -> bytes`` and
``Model.decompress(bytes) -> uint8 pixels``, with ``8 * len(bytes) / pixels`` within a few percent of the estimate.

Tables (host, float64, 16-bit precision: frequencies sum to 65536, every symbol >= 1, last symbol = ESCAPE):
  * y: 64 tables, one per integer scale index k = round(clamp(exp(raw), 0, 63)), sigma_k = SCALE_FN(k)
    (mshyper/models.py:28-32; rounding the index is what TFC's compress() path does), pmf(v) = Phi((v+.5)/s) - Phi((v-.5)/s)
    on |v| <= L_k where the two tails hold < 2^-12 of the mass.
  * z: one table per channel from the deep-factorized prior, pmf(v) = sigmoid(L(v+.5)) - sigmoid(L(v-.5)).
Wire format (little endian): b"SNTC" u16 version | u16 n | u32 H | u32 W | u16 C | u16 hz | u16 wz | u16 h | u16 w |
  u16 group | u16 len_words[streams] (z) | u16 len_words[streams] (y) | z payload | y payload,
  streams = n * ceil(C / group); a stream = [state hi, state lo, words...] of 16-bit words.
The decoder rebuilds mu / scale indexes with the same hyper-synthesis kernels (deterministic, batch-invariant), so
encoder and decoder agree bit for bit.
"""
from __future__ import annotations

import ctypes as C
import math
import struct

import numpy as np
import torch

from . import _capi as capi
from . import ops

PRECISION = 16
TOTAL = 1 << PRECISION
MAGIC = b"SNTC"
VERSION = 1
SCALE_MIN, SCALE_MAX, NUM_SCALES = 0.11, 256.0, 64
SCALE_FACTOR = (math.log(SCALE_MAX) - math.log(SCALE_MIN)) / (NUM_SCALES - 1.0)


def quantize_pmf(pmf, escape_mass):
    """pmf (float64, symbols in value order) + escape -> integer frequencies, each >= 1, summing to 65536."""
    p = np.concatenate([np.maximum(np.asarray(pmf, np.float64), 0.0), [max(float(escape_mass), 0.0)]])
    p = p / p.sum()
    n = len(p)
    if n > TOTAL // 2:
        raise ValueError("table too wide for 16-bit precision")
    f = 1 + np.floor(p * (TOTAL - n)).astype(np.int64)
    f[np.argmax(p)] += TOTAL - int(f.sum())            # the leftover (< n) goes to the most probable symbol
    assert f.min() >= 1 and int(f.sum()) == TOTAL
    return f


def _ndtr(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def normal_tables(tail_mass=2.0 ** -12, max_half_width=4095):
    tabs = []
    for k in range(NUM_SCALES):
        sigma = math.exp(math.log(SCALE_MIN) + SCALE_FACTOR * k)
        L = 0
        while L < max_half_width and 2.0 * _ndtr(-(L + 0.5) / sigma) > tail_mass:
            L += 1
        v = np.arange(-L, L + 1)
        pmf = np.array([_ndtr((t + 0.5) / sigma) - _ndtr((t - 0.5) / sigma) for t in v])
        tabs.append((-L, quantize_pmf(pmf, 2.0 * _ndtr(-(L + 0.5) / sigma))))
    return tabs


def _df_logits(x, mats, biases, factors):
    """Host float64 mirror of tfc.DeepFactorized._logits_cumulative for ONE channel; x: [n]."""
    h = np.asarray(x, np.float64)[None, :]
    nl = len(mats)
    for k in range(nl):
        h = np.logaddexp(0.0, mats[k]) @ h + biases[k][:, None]
        if k < nl - 1:
            h = h + np.tanh(factors[k])[:, None] * np.tanh(h)
    return h[0]


def factorized_tables(prior_weights, num_layers, tail_mass=2.0 ** -12, max_half_width=2047):
    c = prior_weights["prior/matrix_0"].shape[0]
    tabs = []
    for ch in range(c):
        mats = [prior_weights[f"prior/matrix_{k}"][ch].astype(np.float64) for k in range(num_layers)]
        bs = [prior_weights[f"prior/bias_{k}"][ch].astype(np.float64) for k in range(num_layers)]
        fs = [prior_weights[f"prior/factor_{k}"][ch].astype(np.float64) for k in range(num_layers - 1)]
        edges = np.arange(-max_half_width - 0.5, max_half_width + 1.0)           # v - .5 for v = -L..L+1
        cdf = 1.0 / (1.0 + np.exp(-_df_logits(edges, mats, bs, fs)))
        pmf = np.diff(cdf)                                                        # pmf[i] for v = -max + i
        keep = np.nonzero(pmf > tail_mass / 64.0)[0]
        lo, hi = (int(keep[0]), int(keep[-1])) if len(keep) else (max_half_width, max_half_width)
        tabs.append((lo - max_half_width, quantize_pmf(pmf[lo:hi + 1], cdf[lo] + (1.0 - cdf[hi + 1]))))
    return tabs


class DeviceTables:
    """Concatenated CDFs + per-table offsets / sizes / minima on the device."""

    def __init__(self, tabs, device):
        cdfs, off, n, vmin = [], [], [], []
        pos = 0
        for lo, f in tabs:
            cdf = np.concatenate([[0], np.cumsum(f)]).astype(np.uint32)
            off.append(pos)
            n.append(len(f))
            vmin.append(lo)
            cdfs.append(cdf)
            pos += len(cdf)
        self.host = tabs
        self.cdf = torch.from_numpy(np.concatenate(cdfs).astype(np.int64)).to(torch.int32).to(device)   # values <= 65536
        self.off = torch.tensor(off, dtype=torch.int32, device=device)
        self.n = torch.tensor(n, dtype=torch.int32, device=device)
        self.vmin = torch.tensor(vmin, dtype=torch.int32, device=device)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


GROUP = 16      # channels per rANS stream: 6 bytes of overhead per stream vs. decode parallelism


def num_streams(n, c, group=GROUP):
    return n * (-(-c // group))


def rans_encode(values, table_ids, tables: DeviceTables, group=GROUP):
    """values int32 [n, P..., C], table_ids uint16 (int16 storage) same shape -> (payload int16-storage words,
    len_words int64[num_streams])."""
    n, c = values.shape[0], values.shape[-1]
    P = values.numel() // (n * c)
    cap = 2 * P * min(group, c) + 4
    dev = values.device
    ns = num_streams(n, c, group)
    scratch = torch.empty((ns, cap), dtype=torch.int16, device=dev)
    lens = torch.empty((ns,), dtype=torch.int32, device=dev)
    capi.call("sntc_rans_encode", _p(values), _p(table_ids), n, P, c, group, _p(tables.cdf), _p(tables.off), _p(tables.n),
              _p(tables.vmin), cap, _p(scratch), _p(lens), ops._stream())
    lens_h = lens.cpu().numpy().astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens_h)]).astype(np.int64)
    payload = torch.empty((int(offsets[-1]),), dtype=torch.int16, device=dev)
    offs_d = torch.from_numpy(offsets).to(dev)
    capi.call("sntc_rans_compact", _p(scratch), cap, _p(lens), _p(offs_d), ns, _p(payload), ops._stream())
    return payload, lens_h


def rans_decode(payload, lens_h, table_ids, shape, tables: DeviceTables, group=GROUP):
    """-> int32 values of ``shape`` [n, ..., C]; raises on a malformed stream."""
    n, c = shape[0], shape[-1]
    P = int(np.prod(shape)) // (n * c)
    dev = payload.device
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(lens_h)]).astype(np.int64)).to(dev)
    values = torch.empty(tuple(shape), dtype=torch.int32, device=dev)
    bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    capi.call("sntc_rans_decode", _p(payload), _p(offsets), _p(table_ids), n, P, c, group, _p(tables.cdf), _p(tables.off),
              _p(tables.n), _p(tables.vmin), _p(values), _p(bad), ops._stream())
    nbad = int(bad.item())
    if nbad:
        raise capi.SntcError(capi.ERR_BAD_SHAPE, f"bitstream corrupt: {nbad} of {num_streams(n, c, group)} rANS streams did not terminate cleanly")
    return values


def scale_table_ids(hyper):
    n, h, w, c2 = hyper.shape
    tid = torch.empty((n, h, w, c2 // 2), dtype=torch.int16, device=hyper.device)
    capi.call("sntc_scale_table_ids", _p(hyper), n * h * w, c2 // 2, _p(tid), ops._stream())
    return tid


def channel_table_ids(shape, device):
    n, h, w, c = shape
    tid = torch.empty((n, h, w, c), dtype=torch.int16, device=device)
    capi.call("sntc_channel_table_ids", n * h * w, c, _p(tid), ops._stream())
    return tid


def round_to_int(x):
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    capi.call("sntc_round_to_int", _p(x), x.numel(), _p(out), ops._stream())
    return out


def int_to_float(x):
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    capi.call("sntc_int_to_float", _p(x), x.numel(), _p(out), ops._stream())
    return out


class Codec:
    """compress / decompress for a mean-scale hyperprior ``Model``."""

    def __init__(self, model):
        self.m = model
        dev = model.device
        nl = len(model._prior_num_filters) + 1
        with torch.cuda.device(dev):
            self.y_tables = DeviceTables(normal_tables(), dev)
            self.z_tables = DeviceTables(factorized_tables(model._prior_weights, nl), dev)

    def compress(self, x) -> bytes:
        m = self.m
        x = m._as_device_images(x)
        n, H, W, _ = x.shape
        with torch.cuda.device(m.device):
            lat = m.infer_latent_rvs(x)
            z, y = lat.uq[0].loc, lat.uq[1].loc
            zi = round_to_int(z)
            z_hat = int_to_float(zi)
            hyper = m._hyper_synthesis(z_hat)
            _, _, sym = ops.entropy_scale_normal(y, hyper, want_symbols=True)
            zp, zl = rans_encode(zi, channel_table_ids(z.shape, m.device), self.z_tables)
            yp, yl = rans_encode(sym, scale_table_ids(hyper), self.y_tables)
            zb, yb = zp.cpu().numpy().tobytes(), yp.cpu().numpy().tobytes()
        if max(int(zl.max()), int(yl.max())) > 0xFFFF:
            raise capi.SntcError(capi.ERR_UNSUPPORTED, "a rANS stream exceeds 65535 words; lower entropy_coding.GROUP")
        head = MAGIC + struct.pack("<HHIIHHHHHH", VERSION, n, H, W, y.shape[-1], z.shape[1], z.shape[2], y.shape[1], y.shape[2], GROUP)
        return head + zl.astype("<u2").tobytes() + yl.astype("<u2").tobytes() + zb + yb

    def decompress(self, blob: bytes):
        m = self.m
        if blob[:4] != MAGIC:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "not an SNTC bitstream")
        ver, n, H, W, c, hz, wz, h, w, group = struct.unpack_from("<HHIIHHHHHH", blob, 4)
        if ver != VERSION:
            raise capi.SntcError(capi.ERR_UNSUPPORTED, f"bitstream version {ver}")
        pos = 4 + struct.calcsize("<HHIIHHHHHH")
        ns = num_streams(n, c, group)
        if len(blob) < pos + 4 * ns:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "bitstream truncated")
        zl = np.frombuffer(blob, "<u2", ns, pos).astype(np.int64)
        yl = np.frombuffer(blob, "<u2", ns, pos + 2 * ns).astype(np.int64)
        pos += 4 * ns
        zw, yw = int(zl.sum()), int(yl.sum())
        if len(blob) != pos + 2 * (zw + yw):
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "bitstream truncated")
        dev = m.device
        with torch.cuda.device(dev):
            zp = torch.from_numpy(np.frombuffer(blob, "<i2", zw, pos).copy()).to(dev)
            yp = torch.from_numpy(np.frombuffer(blob, "<i2", yw, pos + 2 * zw).copy()).to(dev)
            zi = rans_decode(zp, zl, channel_table_ids((n, hz, wz, c), dev), (n, hz, wz, c), self.z_tables, group)
            hyper = m._hyper_synthesis(int_to_float(zi))
            sym = rans_decode(yp, yl, scale_table_ids(hyper), (n, h, w, c), self.y_tables, group)
            y_hat = ops.dequant_scale_normal(sym, hyper)
            return ops.to_pixels(m._synthesis(y_hat), H, W)

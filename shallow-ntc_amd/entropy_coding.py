"""Bitstream for the mean-scale hyperprior codec: integer CDF tables + rANS on the GPU.

The reference never produces a bitstream (``compression=False`` everywhere, mshyper/models.py:246-251): its
bpp is -sum log2 p.  This module makes ``decode`` a real codec: ``Model.compress(x) -> bytes`` and
``Model.decompress(bytes) -> uint8 pixels``, with ``8 * len(bytes) / pixels`` within a few percent of the estimate.

Tables (host, float64, 16-bit precision: frequencies sum to 65536, every symbol >= 1, last symbol = ESCAPE):
  * y: 64 tables, one per integer scale index k = round(clamp(exp(raw), 0, 63)), sigma_k = SCALE_FN(k)
    (mshyper/models.py:28-32; rounding the index is what TFC's compress() path does), pmf(v) = Phi((v+.5)/s) - Phi((v-.5)/s)
    on |v| <= L_k = the symbols with pmf >= 2^-17; rarer values are escaped.
  * z: one table per channel from the deep-factorized prior, pmf(v) = sigmoid(L(v+.5)) - sigmoid(L(v-.5)).
Wire format v3 (little endian): b"SNTC" u16 version (low byte 3; high byte = arithmetic of the transforms that rebuild mu / sigma:
  0 fp32, 1 bf16x3 -- a decoder of the other arithmetic refuses the stream instead of decoding garbage) | u16 n | u32 H | u32 W | u16 C | u16 Cz | u16 hz | u16 wz | u16 h | u16 w |
  u16 segments_z | u16 segments_y | u8 lanes_z | u8 lanes_y | u32 len_words[n * segments_z] | u32 len_words[n * segments_y] |
  z payload | y payload; a stream (one per image and segment) = the lane states (8 .. 64 of them, fewer on short streams)
  + the interleaved 16-bit words (csrc/rans.hip).
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
VERSION = 3
ARITH = {"fp32": 0, "bf16x3": 1}          # Model(precision=...): high byte of the version word
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


def normal_tables(min_pmf=2.0 ** -17, max_half_width=4095):
    """64 tables; table k spans |v| <= L_k, the symbols whose probability is worth a frequency count of its own
    (pmf >= 2^-17, i.e. >= 1/2 count at 16-bit precision).  Rarer values go through ESCAPE (~32 bits against an
    ideal > 17): the extra cost is bounded by 15 bits x 2^-17 per symbol."""
    tabs = []
    for k in range(NUM_SCALES):
        sigma = math.exp(math.log(SCALE_MIN) + SCALE_FACTOR * k)
        pmf_at = lambda t: _ndtr((t + 0.5) / sigma) - _ndtr((t - 0.5) / sigma) if t <= 0 else _ndtr(-(t - 0.5) / sigma) - _ndtr(-(t + 0.5) / sigma)
        L = 0
        while L < max_half_width and pmf_at(L + 1) >= min_pmf:
            L += 1
        v = np.arange(-L, L + 1)
        pmf = np.array([pmf_at(int(t)) for t in v])
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


def decoder_entries(tabs):
    """(cdf[s] << 16) | (freq[s] - 1) for every symbol of every table, three 0xffffffff after each table, padded to a multiple
    of four entries."""
    out = []
    for _, f in tabs:
        f = np.asarray(f, np.int64)
        cdf = np.concatenate([[0], np.cumsum(f)[:-1]])
        assert len(f) >= 2 and f.min() >= 1 and f.max() <= 65535
        out.append(((cdf << 16) | (f - 1)).astype(np.uint32))
        out.append(np.full(3, 0xFFFFFFFF, np.uint32))
    flat = np.concatenate(out)
    pad = -len(flat) % 4
    return np.concatenate([flat, np.full(pad, 0xFFFFFFFF, np.uint32)]) if pad else flat


def start_tables(cdfs, bits):
    """Per table the decoder's start table: entry b of 2^bits = the largest symbol s with cdf[s] <= b << (16 - bits).
    -> (uint16 entries of all tables, padded to a multiple of eight; uint32 (offset << 5) | bits per table)."""
    luts, lmeta, pos = [], [], 0
    for cdf, b in zip(cdfs, bits):
        base = np.arange(1 << b, dtype=np.int64) << (16 - b)
        lut = np.searchsorted(cdf.astype(np.int64), base, side="right") - 1
        assert lut.min() >= 0 and lut.max() < len(cdf)
        luts.append(lut.astype(np.uint16))
        lmeta.append((pos << 5) | b)
        pos += len(lut)
    flat = np.concatenate(luts)
    pad = -len(flat) % 8
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, np.uint16)]).astype(np.uint16)
    return flat, np.asarray(lmeta, np.uint32)


class DeviceTables:
    """Concatenated uint16 CDFs (cdf[n] = 65536 implicit) + packed per-table descriptors on the device."""

    def __init__(self, tabs, device):
        cdfs, meta = [], []
        pos = 0
        for lo, f in tabs:
            cdf = np.concatenate([[0], np.cumsum(f)[:-1]]).astype(np.uint16)     # cdf of symbols 0..n-1
            if not -32768 <= lo <= 32767 or len(f) > 32768:
                raise ValueError("table outside the 16-bit descriptor range")
            meta.append((pos, (len(f) << 16) | (lo & 0xFFFF)))
            cdfs.append(cdf)
            pos += len(cdf)
        flat = np.concatenate(cdfs)
        if len(flat) % 2:
            flat = np.concatenate([flat, [0]]).astype(np.uint16)                  # kernels copy 32 bits at a time
        self.host = tabs
        self.ntables, self.total = len(tabs), pos
        self.cdf = torch.from_numpy(flat.view(np.int16).copy()).to(device)
        self.meta = torch.from_numpy(np.asarray(meta, np.uint32).view(np.int32).copy()).to(device)
        # the decoder's own view of the tables (include/sntc.h, sntc_rans_decode): packed (start, frequency - 1) entries with
        # three sentinels per table, and start tables of about one entry per symbol -- a table of n symbols gets 2^ceil(log2 n)
        # buckets, one bit more where the CU's LDS has the room, fewer where it has not
        budget = int(capi.load().sntc_rans_lut_budget(self.ntables, self.total))
        self.dec, self.lut, self.lut_meta, self.lut_total, self.lut_bits = None, None, None, 0, None
        for extra in (1, 0, -1, -2, -3, -4, -5, -6):
            bits = [min(16, max(0, int(np.ceil(np.log2(max(len(f), 1)))) + extra)) for _, f in tabs]
            if -(-sum(1 << b for b in bits) // 8) * 8 <= budget:
                lut, lmeta = start_tables(cdfs, bits)
                dec = decoder_entries(tabs)
                self.lut_bits, self.lut_total = bits, len(lut)
                self.dec = torch.from_numpy(dec.view(np.int32).copy()).to(device)
                self.lut = torch.from_numpy(lut.view(np.int16).copy()).to(device)
                self.lut_meta = torch.from_numpy(lmeta.view(np.int32).copy()).to(device)
                break


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


PIPELINE_BLOBS = True           # decompress_many: blobs pipelined largest first (False: round 4's two-phase schedule; same pixels)
USE_START_TABLES = True         # decoder: its own tables in LDS, DeviceTables.dec / .lut (False: binary search of cdf; same values)

ELEMS_PER_SEGMENT = 1 << 18     # one rANS stream (= one wave of coding parallelism, 256 bytes of flushed lane states) per
                                # this many latent elements: ~0.008 bit / element of overhead; a Kodak image = 1 z + 2 y streams


def _segments(elems, segments=None):
    if segments is None:
        segments = -(-elems // ELEMS_PER_SEGMENT)
    return max(1, min(int(segments), -(-elems // 64)))


def _lanes(elems_per_stream, lanes=None):
    """Lane states flushed per stream (4 bytes each): all 64 on long streams, fewer on short ones (the hyper-latents of a
    small image would otherwise pay 256 bytes of states for a few hundred bytes of payload)."""
    if lanes is not None:
        return int(lanes)
    return 64 if elems_per_stream >= 16384 else 32 if elems_per_stream >= 6144 else 16 if elems_per_stream >= 2048 else 8


def rans_encode_launch(values, table_ids, tables: DeviceTables, segments=None, lanes=None):
    """The device half of ``rans_encode``: every stream coded into its own ``cap``-word scratch row, no host synchronisation.
    -> (scratch int16 [streams, cap], lens int32 [streams]) for ``rans_encode_finish``."""
    n = values.shape[0]
    E = values.numel() // n
    segments = _segments(E, segments)
    lanes = _lanes(-(-E // segments), lanes)
    cap = int(capi.load().sntc_rans_cap_words(E, segments))
    dev = values.device
    ns = n * segments
    scratch = torch.empty((ns, cap), dtype=torch.int16, device=dev)
    lens = torch.empty((ns,), dtype=torch.int32, device=dev)
    capi.call("sntc_rans_encode", _p(values), _p(table_ids), n, E, segments, lanes, _p(tables.cdf), _p(tables.meta), tables.ntables,
              tables.total, cap, _p(scratch), _p(lens), ops._stream())
    return scratch, lens


def rans_encode_finish(scratch, lens, lens_h):
    """The streams of ``rans_encode_launch`` packed back to back: ``lens_h`` = ``lens`` on the host (int64).  -> payload (device)."""
    dev = scratch.device
    offsets = np.concatenate([[0], np.cumsum(lens_h)]).astype(np.int64)
    payload = torch.empty((int(offsets[-1]),), dtype=torch.int16, device=dev)
    offs_d = torch.from_numpy(offsets).to(dev)
    capi.call("sntc_rans_compact", _p(scratch), scratch.shape[1], _p(lens), _p(offs_d), scratch.shape[0], _p(payload), ops._stream())
    return payload


def rans_encode(values, table_ids, tables: DeviceTables, segments=None, lanes=None):
    """values int32 [n, ...], table_ids uint16 (int16 storage) same shape -> (payload int16-storage words on the
    device, len_words int64[n * segments])."""
    scratch, lens = rans_encode_launch(values, table_ids, tables, segments, lanes)
    lens_h = lens.cpu().numpy().astype(np.int64)
    return rans_encode_finish(scratch, lens, lens_h), lens_h


def rans_decode(payload, lens_h, table_ids, shape, tables: DeviceTables, segments=None, lanes=None, bad=None, offsets=None):
    """-> int32 values of ``shape`` [n, ...]; raises on a malformed stream.  ``bad`` (an int32 device tensor [1]): count the
    streams that did not terminate cleanly there instead of reading the count back here -- the caller checks it where it
    synchronises anyway (a decoder that keeps several batches in flight)."""
    n = shape[0]
    E = int(np.prod(shape)) // n
    segments = _segments(E, segments)
    lanes = _lanes(-(-E // segments), lanes)
    if len(lens_h) != n * segments:
        raise capi.SntcError(capi.ERR_BAD_SHAPE, "stream count does not match the image / segment counts")
    dev = payload.device
    if offsets is None:        # ``offsets``: the exclusive prefix sum of lens_h already on the device (int64 [streams + 1])
        offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(lens_h)]).astype(np.int64)).to(dev)
    values = torch.empty(tuple(shape), dtype=torch.int32, device=dev)
    deferred = bad is not None
    if not deferred:
        bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    fast = USE_START_TABLES and tables.dec is not None
    capi.call("sntc_rans_decode", _p(payload), _p(offsets), _p(table_ids), n, E, segments, lanes, _p(tables.cdf), _p(tables.meta),
              tables.ntables, tables.total, _p(tables.dec if fast else None), _p(tables.lut if fast else None),
              _p(tables.lut_meta if fast else None), tables.lut_total if fast else 0, _p(values), _p(bad), ops._stream())
    if not deferred:
        nbad = int(bad.item())
        if nbad:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, f"bitstream corrupt: {nbad} of {n * segments} rANS streams did not terminate cleanly")
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

    HEAD = "<HHIIHHHHHHHHBB"
    MAX_IMAGES, MAX_SIDE = 4096, 1 << 16

    def __init__(self, model):
        self.m = model
        dev = model.device
        nl = len(model._prior_num_filters) + 1
        with torch.cuda.device(dev):
            self.y_tables = DeviceTables(normal_tables(), dev)
            self.z_tables = DeviceTables(factorized_tables(model._prior_weights, nl), dev)

    def latent_shapes(self, H, W):
        """(C, Cz, hz, wz, h, w) of this model's latents for an H x W image (pad to the downsample factor, then the
        transforms' own shape arithmetic)."""
        m = self.m
        f = m.downsample_factor
        hp, wp = -(-H // f) * f, -(-W // f) * f
        h, w = m._analysis.out_hw(hp, wp)
        hz, wz = m._hyper_analysis.out_hw(h, w)
        return (m._bottleneck_size, m._hyper_bottleneck_size, hz, wz, h, w)

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
            sz, sy = _segments(zi[0].numel()), _segments(sym[0].numel())
            lz, ly = _lanes(-(-zi[0].numel() // sz)), _lanes(-(-sym[0].numel() // sy))
            zp, zl = rans_encode(zi, channel_table_ids(z.shape, m.device), self.z_tables, sz, lz)
            yp, yl = rans_encode(sym, scale_table_ids(hyper), self.y_tables, sy, ly)
            zb, yb = zp.cpu().numpy().tobytes(), yp.cpu().numpy().tobytes()
            ops.check_conv_status()       # the copies synchronised the stream: a flagged stream-K launch raises here, not a wrong file
        head = MAGIC + struct.pack(self.HEAD, VERSION | (ARITH[m._precision] << 8), n, H, W, y.shape[-1], z.shape[-1], z.shape[1], z.shape[2], y.shape[1], y.shape[2],
                                   sz, sy, lz, ly)
        return head + zl.astype("<u4").tobytes() + yl.astype("<u4").tobytes() + zb + yb

    def compress_many(self, xs):
        """``compress`` for several batches (e.g. one per image size of a set) -> their bitstreams, in order, byte for byte what one
        ``compress`` per batch returns.  The batches' transforms and entropy-coding launches run side by side on the library's
        side streams; the stream lengths of ALL batches come back in one copy, the packed payloads in another -- two host
        synchronisations for the set instead of four per batch."""
        m = self.m
        xs = [m._as_device_images(x) for x in xs]
        if not xs:
            return []
        with torch.cuda.device(m.device):
            main = torch.cuda.current_stream()
            side = ops.side_streams(len(xs), m.device) if len(xs) > 1 and not torch.cuda.is_current_stream_capturing() else [main] * len(xs)
            jobs = []
            for st, x in zip(side, xs):
                if st is not main:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    lat = m.infer_latent_rvs(x)
                    z, y = lat.uq[0].loc, lat.uq[1].loc
                    zi = round_to_int(z)
                    hyper = m._hyper_synthesis(int_to_float(zi))
                    _, _, sym = ops.entropy_scale_normal(y, hyper, want_symbols=True)
                    sz, sy = _segments(zi[0].numel()), _segments(sym[0].numel())
                    lz, ly = _lanes(-(-zi[0].numel() // sz)), _lanes(-(-sym[0].numel() // sy))
                    zs, zlen = rans_encode_launch(zi, channel_table_ids(z.shape, m.device), self.z_tables, sz, lz)
                    ys, ylen = rans_encode_launch(sym, scale_table_ids(hyper), self.y_tables, sy, ly)
                jobs.append(dict(x=x, z=z, y=y, sz=sz, sy=sy, lz=lz, ly=ly, zs=zs, zlen=zlen, ys=ys, ylen=ylen, st=st))
            for j in jobs:
                if j["st"] is not main:
                    main.wait_stream(j["st"])
                    for t in (j["zs"], j["zlen"], j["ys"], j["ylen"]):
                        t.record_stream(main)
            lens_h = torch.cat([t for j in jobs for t in (j["zlen"], j["ylen"])]).cpu().numpy().astype(np.int64)     # read-back 1 of 2
            pays, o = [], 0
            for j in jobs:
                nz, ny = j["zlen"].numel(), j["ylen"].numel()
                j["zl"], j["yl"] = lens_h[o:o + nz], lens_h[o + nz:o + nz + ny]
                o += nz + ny
                pays += [rans_encode_finish(j["zs"], j["zlen"], j["zl"]), rans_encode_finish(j["ys"], j["ylen"], j["yl"])]
            words = torch.cat(pays).cpu().numpy()                                                                    # read-back 2 of 2
            ops.check_conv_status()       # the copies synchronised the stream: a flagged stream-K launch raises here, not a wrong file
        out, o = [], 0
        for j in jobs:
            n, H, W, _ = j["x"].shape
            z, y = j["z"], j["y"]
            zw, yw = int(j["zl"].sum()), int(j["yl"].sum())
            head = MAGIC + struct.pack(self.HEAD, VERSION | (ARITH[m._precision] << 8), n, H, W, y.shape[-1], z.shape[-1], z.shape[1], z.shape[2],
                                       y.shape[1], y.shape[2], j["sz"], j["sy"], j["lz"], j["ly"])
            out.append(head + j["zl"].astype("<u4").tobytes() + j["yl"].astype("<u4").tobytes() + words[o:o + zw + yw].tobytes())
            o += zw + yw
        return out

    def _parse(self, blob: bytes):
        """Header and stream lengths of one blob, checked against THIS model: nothing later trusts the header."""
        m = self.m
        if blob[:4] != MAGIC:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "not an SNTC bitstream")
        pos = 4 + struct.calcsize(self.HEAD)
        if len(blob) < pos:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "bitstream truncated")
        ver, n, H, W, c, cz, hz, wz, h, w, sz, sy, lz, ly = struct.unpack_from(self.HEAD, blob, 4)
        arith, ver = ver >> 8, ver & 0xff
        if ver != VERSION:
            raise capi.SntcError(capi.ERR_UNSUPPORTED, f"bitstream version {ver}")
        if arith != ARITH[m._precision]:
            names = {v: k for k, v in ARITH.items()}
            raise capi.SntcError(capi.ERR_UNSUPPORTED, f"bitstream was written by a {names.get(arith, arith)!r} model, this model "
                                 f"computes in {m._precision!r}: mu / sigma would not be reproduced bit for bit")
        # Every dimension is recomputed from (H, W) and THIS model, so a corrupt or crafted blob cannot size an allocation or
        # index a table-id tensor beyond what the model itself would produce.
        if not (1 <= n <= self.MAX_IMAGES and 1 <= H <= self.MAX_SIDE and 1 <= W <= self.MAX_SIDE):
            raise capi.SntcError(capi.ERR_BAD_SHAPE, f"bitstream header: implausible batch / image size n={n} H={H} W={W}")
        want = self.latent_shapes(H, W)
        if (c, cz, hz, wz, h, w) != want:
            raise capi.SntcError(capi.ERR_BAD_SHAPE, f"bitstream header (C, Cz, hz, wz, h, w) = {(c, cz, hz, wz, h, w)} does not match "
                                 f"this model's latents for a {H} x {W} image: {want}")
        ez, ey = hz * wz * cz, h * w * c
        if (sz, sy) != (_segments(ez), _segments(ey)) or (lz, ly) != (_lanes(-(-ez // sz)), _lanes(-(-ey // sy))):
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "bitstream header: segment / lane counts do not match the latent sizes")
        nz, ny = n * sz, n * sy
        if len(blob) < pos + 4 * (nz + ny):
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "bitstream truncated")
        zl = np.frombuffer(blob, "<u4", nz, pos).astype(np.int64)
        yl = np.frombuffer(blob, "<u4", ny, pos + 4 * nz).astype(np.int64)
        pos += 4 * (nz + ny)
        zw, yw = int(zl.sum()), int(yl.sum())
        if len(blob) != pos + 2 * (zw + yw):
            raise capi.SntcError(capi.ERR_BAD_SHAPE, "bitstream truncated")
        return dict(n=n, H=H, W=W, c=c, cz=cz, hz=hz, wz=wz, h=h, w=w, sz=sz, sy=sy, lz=lz, ly=ly, zl=zl, yl=yl, pos=pos, zw=zw, yw=yw)

    def decompress(self, blob: bytes):
        return self.decompress_many([blob])[0]

    def decompress_many(self, blobs):
        """Several bitstreams (e.g. one per batch shape of a set) -> their pixel batches, in order.  An entropy-decoding launch
        is a handful of lone waves whose time is a latency (one wave per stream, ~0.4 us per step of 64 symbols whatever the
        batch), so the launches of ALL blobs' hyper-latents run side by side on streams of their own; then the blobs are
        pipelined, largest first: blob k's latents decode on its stream while the caller's stream runs blob k + 1's
        hyper-synthesis and, later, blob k - 1's synthesis.  A decoding wave holds ~100 KB of tables in its CU's LDS and a
        stream-K convolution needs every one of its workers resident (round 4 measured what happens when they meet: groups of one
        blob's images pipelined against each other's stream-K convolutions, 9.5 -> 10.9 / 12.7 / 17.6 ms with 2 / 3 / 4 groups),
        so only the FIRST hyper-synthesis -- which meets no decoding wave -- keeps stream-K; the convolutions that may run beside
        decoding waves take the static schedules (``ops.static_schedules``; same bits)."""
        m = self.m
        if not blobs:
            return []
        heads = [self._parse(b) for b in blobs]
        dev = m.device
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream()
            side = self._side_streams(len(blobs)) if len(blobs) > 1 and not torch.cuda.is_current_stream_capturing() else [main] * len(blobs)
            # one counter per entropy-decoding launch (the launch zeroes its own), ONE read-back for everything at the end: the
            # chain z symbols -> hyper-synthesis -> y symbols -> synthesis is enqueued without the host waiting in between
            bad = torch.zeros((2 * len(blobs),), dtype=torch.int32, device=dev)
            # ONE upload for every blob's words and stream offsets (int64 offsets first, then the 16-bit words: both aligned)
            offs = [np.concatenate([[0], np.cumsum(hd[k])]).astype(np.int64) for hd in heads for k in ("zl", "yl")]
            words = [np.frombuffer(b, "<i2", hd["zw"] + hd["yw"], hd["pos"]) for b, hd in zip(blobs, heads)]
            noff = sum(len(o) for o in offs)
            host = np.empty(8 * noff + 2 * sum(len(w) for w in words), np.uint8)
            host[:8 * noff].view(np.int64)[:] = np.concatenate(offs)
            host[8 * noff:].view(np.int16)[:] = np.concatenate(words)
            up = torch.from_numpy(host).to(dev)
            off_d, word_d = up[:8 * noff].view(torch.int64), up[8 * noff:].view(torch.int16)
            pay, offd = [], []
            o = wpos = 0
            for hd in heads:
                nz, ny = len(hd["zl"]) + 1, len(hd["yl"]) + 1
                offd.append((off_d[o:o + nz], off_d[o + nz:o + nz + ny]))
                o += nz + ny
                pay.append((word_d[wpos:wpos + hd["zw"]], word_d[wpos + hd["zw"]:wpos + hd["zw"] + hd["yw"]]))
                wpos += hd["zw"] + hd["yw"]

            def side_by_side(jobs):
                outs = []
                for st, job in zip(side, jobs):
                    if st is not main:
                        st.wait_stream(main)
                    with torch.cuda.stream(st):
                        outs.append(job())
                for st, out in zip(side, outs):
                    if st is not main:
                        main.wait_stream(st)
                        out.record_stream(main)
                return outs

            zis = side_by_side([
                (lambda k=k, hd=hd: rans_decode(pay[k][0], hd["zl"], channel_table_ids((hd["n"], hd["hz"], hd["wz"], hd["cz"]), dev),
                                                (hd["n"], hd["hz"], hd["wz"], hd["cz"]), self.z_tables, hd["sz"], hd["lz"], bad=bad[2 * k:2 * k + 1],
                                                offsets=offd[k][0]))
                for k, hd in enumerate(heads)])
            # From here on the blobs are PIPELINED, largest first: a blob's latents decode on its side stream while the caller's
            # stream runs the next blob's hyper-synthesis, and its synthesis runs while the later blobs' latents decode.  The first
            # blob's hyper-synthesis meets no decoding wave and keeps its stream-K launches; every convolution after it may run
            # beside decoding waves and takes the static schedules (ops.static_schedules: same bits).
            piped = PIPELINE_BLOBS and len(heads) > 1 and side[0] is not main
            order = sorted(range(len(heads)), key=lambda k: -(heads[k]["n"] * heads[k]["h"] * heads[k]["w"])) if piped else list(range(len(heads)))
            hypers, syms, tidl = [None] * len(heads), [None] * len(heads), [None] * len(heads)

            def launch_latents(k):
                hd, st = heads[k], side[k]
                if st is not main:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    syms[k] = rans_decode(pay[k][1], hd["yl"], tidl[k], (hd["n"], hd["h"], hd["w"], hd["c"]), self.y_tables, hd["sy"], hd["ly"],
                                          bad=bad[2 * k + 1:2 * k + 2], offsets=offd[k][1])
                if st is not main:
                    tidl[k].record_stream(st)

            for i, k in enumerate(order):
                hd = heads[k]
                with ops.static_schedules(piped and i > 0):
                    hyper = m._hyper_synthesis(int_to_float(zis[k]))
                if tuple(hyper.shape) != (hd["n"], hd["h"], hd["w"], 2 * hd["c"]):
                    raise capi.SntcError(capi.ERR_BAD_SHAPE, f"hyper-synthesis output {tuple(hyper.shape)} does not match the latents "
                                         f"{(hd['n'], hd['h'], hd['w'], hd['c'])}")
                hypers[k] = hyper
                tidl[k] = scale_table_ids(hyper)
                if piped:
                    launch_latents(k)
            if not piped:                    # A/B (PIPELINE_BLOBS = False), round 4's schedule: every hyper-synthesis, then every blob's
                for k in order:              # latents side by side, then the syntheses
                    launch_latents(k)
            out = [None] * len(heads)
            if not piped:                    # A/B (PIPELINE_BLOBS = False): every blob's latents first, then the syntheses
                for k in order:
                    if side[k] is not main:
                        main.wait_stream(side[k])
            for i, k in enumerate(order):
                hd, st = heads[k], side[k]
                if st is not main:
                    main.wait_stream(st)
                    syms[k].record_stream(main)
                with ops.static_schedules(piped and i + 1 < len(order)):
                    y_hat = ops.dequant_split3(syms[k], hypers[k]) if m._synthesis.takes_s3(hd["h"], hd["w"]) else ops.dequant_scale_normal(syms[k], hypers[k])
                    out[k] = m._pixels(y_hat, (hd["H"], hd["W"]))
            nbad = int(bad.sum().item())                           # synchronises the stream
            if nbad:
                total = sum(hd["n"] * (hd["sz"] + hd["sy"]) for hd in heads)
                raise capi.SntcError(capi.ERR_BAD_SHAPE, f"bitstream corrupt: {nbad} of {total} rANS streams did not terminate cleanly")
            ops.check_conv_status()       # wrong pixels never leave without an error
            return out

    def _side_streams(self, count):
        return ops.side_streams(count, self.m.device)

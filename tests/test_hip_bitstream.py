"""rANS bitstream on the GPU (-m gpu): word-exact against the pure-Python restatement of the
stream format, encode -> decode round trips in the integer domain, rate vs the estimate, corruption detection."""
import numpy as np
import pytest
import torch

from oracle import rans_np

pytestmark = pytest.mark.gpu


def test_tables_are_valid_distributions(dev):
    from shallow_ntc_amd import entropy_coding as ec
    tabs = ec.normal_tables()
    assert len(tabs) == 64
    for k, (lo, f) in enumerate(tabs):
        assert int(f.sum()) == 65536 and f.min() >= 1 and lo == -(len(f) - 2) // 2
    assert len(tabs[0][1]) == 2 and tabs[63][0] < -600              # sigma 0.11: {0, ESCAPE}; sigma 256: wide support
    f = ec.quantize_pmf([0.5, 0.25, 0.25], 0.0)
    assert int(f.sum()) == 65536 and f[3] == 1 and f[0] > f[1] == f[2] and abs(int(f[0]) - 32768) < 8


def test_stream_words_match_python_restatement(dev):
    from shallow_ntc_amd import entropy_coding as ec
    rng = np.random.default_rng(0)
    tabs = ec.normal_tables()
    dt = ec.DeviceTables(tabs, dev)
    n, P, c = 2, 101, 5
    tids = rng.integers(0, 64, size=(n, P, c)).astype(np.int16)
    sig = np.array([0.11 * np.exp(ec.SCALE_FACTOR * k) for k in range(64)])
    vals = np.rint(rng.laplace(0, 1, size=(n, P, c)) * sig[tids] * 1.5).astype(np.int32)
    vals[0, 3, 1] = 20000          # escapes
    vals[1, 0, 0] = -31000
    vals[1, 100, 4] = 32767
    E = P * c                               # 505 elements: 8 steps of 64 lanes, the last one ragged
    for segs, lanes in ((1, 64), (2, 32), (3, 8), (1, 16)):
        eseg = -(-(-(-E // segs)) // 64) * 64
        payload, lens = ec.rans_encode(torch.from_numpy(vals).to(dev), torch.from_numpy(tids).to(dev), dt, segs, lanes)
        words = payload.cpu().numpy().view(np.uint16)
        off = np.concatenate([[0], np.cumsum(lens)])
        assert len(lens) == n * segs
        for b in range(n):
            for g in range(segs):
                s = b * segs + g
                sl = slice(g * eseg, min(E, (g + 1) * eseg))
                v, t = vals[b].ravel()[sl], tids[b].ravel()[sl]
                ref = rans_np.encode_stream(v, t, tabs, lanes)
                got = words[off[s]:off[s + 1]].tolist()
                assert got == ref, (segs, lanes, b, g)
                assert rans_np.decode_stream(got, t, tabs, lanes) == v.tolist()
        back = ec.rans_decode(payload, lens, torch.from_numpy(tids).to(dev), (n, P, c), dt, segs, lanes)
        np.testing.assert_array_equal(back.cpu().numpy(), vals)
    payload, lens = ec.rans_encode(torch.from_numpy(vals).to(dev), torch.from_numpy(tids).to(dev), dt)
    # corruption: flip one payload word -> decode must refuse
    bad = payload.clone()
    bad[60] ^= 0x0100
    from shallow_ntc_amd import _capi
    with pytest.raises(_capi.SntcError, match="corrupt"):
        ec.rans_decode(bad, lens, torch.from_numpy(tids).to(dev), (n, P, c), dt)


def test_start_tables_do_not_change_a_decoded_value(dev, monkeypatch):
    """The decoder's start tables (DeviceTables.lut) only shorten the symbol search: same values with and without them, on every
    table incl. the thin tails of the widest ones (where a bucket spans dozens of symbols), on the per-channel tables of a
    factorized prior, and the same refusal of a corrupt stream."""
    from shallow_ntc_amd import _capi
    from shallow_ntc_amd import entropy_coding as ec
    rng = np.random.default_rng(5)
    tabs = ec.normal_tables()
    dt = ec.DeviceTables(tabs, dev)
    assert dt.lut is not None and dt.lut_total <= int(_capi.load().sntc_rans_lut_budget(dt.ntables, dt.total))
    n, P, c = 3, 700, 16
    tids = rng.integers(0, 64, size=(n, P, c)).astype(np.int16)
    tids[1] = 63                                               # one image entirely on the widest table
    sig = np.array([0.11 * np.exp(ec.SCALE_FACTOR * k) for k in range(64)])
    vals = np.rint(rng.standard_normal((n, P, c)) * sig[tids] * np.where(rng.random((n, P, c)) < 0.2, 3.5, 1.0)).astype(np.int32)
    vals[2, :40, 0] = rng.integers(-30000, 30000, 40)         # escapes
    payload, lens = ec.rans_encode(torch.from_numpy(vals).to(dev), torch.from_numpy(tids).to(dev), dt)
    outs = []
    for use in (True, False):
        monkeypatch.setattr(ec, "USE_START_TABLES", use)
        outs.append(ec.rans_decode(payload, lens, torch.from_numpy(tids).to(dev), (n, P, c), dt))
        bad = payload.clone()
        bad[200] ^= 0x0440
        with pytest.raises(_capi.SntcError, match="corrupt"):
            ec.rans_decode(bad, lens, torch.from_numpy(tids).to(dev), (n, P, c), dt)
    np.testing.assert_array_equal(outs[0].cpu().numpy(), vals)
    assert torch.equal(outs[0], outs[1])
    # the three decoder-table arguments come together, and within the LDS budget
    import ctypes as C
    P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(dev)
    out, badc = torch.empty((n, 700, 16), dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)
    common = (P(payload), P(offs), P(torch.from_numpy(tids).to(dev)), n, 700 * 16, 1, 64, P(dt.cdf), P(dt.meta), dt.ntables, dt.total)
    for dec, lut, lmeta, entries in ((dt.dec, None, dt.lut_meta, dt.lut_total), (None, dt.lut, dt.lut_meta, dt.lut_total),
                                     (dt.dec, dt.lut, dt.lut_meta, dt.lut_total + 8 * 8192), (dt.dec, dt.lut, dt.lut_meta, dt.lut_total - 1)):
        with pytest.raises(_capi.SntcError):
            _capi.call("sntc_rans_decode", *common, P(dec), P(lut), P(lmeta), entries, P(out), P(badc), None)
    # every slot of every table: the start symbol is at or below the slot's symbol, never above
    for (lo, f), bits, lm in zip(tabs, dt.lut_bits, dt.lut_meta.cpu().numpy().view(np.uint32)):
        cdf = np.concatenate([[0], np.cumsum(f)[:-1]])
        assert lm & 31 == bits
        lut = dt.lut.cpu().numpy().view(np.uint16)[(lm >> 5):(lm >> 5) + (1 << bits)]
        slots = np.arange(0, 65536, 7)
        sym = np.searchsorted(cdf, slots, side="right") - 1
        start = lut[slots >> (16 - bits)]
        assert (start <= sym).all() and (cdf[start] <= (slots >> (16 - bits) << (16 - bits))).all()


def test_codec_round_trip_and_rate(dev):
    """compress -> bytes -> decompress == decode(encode(x)) bit for bit; the coded size is within a few percent
    of the estimated rate (integer scale table + 16-bit frequencies + escapes + per-stream headers)."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.02))
    w = dict(model.get_weights())
    rng = np.random.default_rng(0)
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[320:] = rng.uniform(-1.0, 2.5, size=320)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    model.set_weights(w)
    x = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(2, 512, 768, seed=8))).to(dev)
    blob = model.compress(x)
    assert blob[:4] == b"SNTC"
    px = model.decompress(blob)
    z_hat, sym, bits_z, bits_y = model.encode(x)
    ref = model.decode(z_hat, sym, (512, 768))
    assert torch.equal(px, ref)
    est_bits = float(bits_z.sum() + bits_y.sum())
    real_bits = 8.0 * len(blob)
    assert 0.97 * est_bits < real_bits < 1.08 * est_bits, (real_bits, est_bits)
    # single images decode identically from their own streams (batch invariance of the whole chain)
    blob0 = model.compress(x[:1].contiguous())
    px0 = model.decompress(blob0)
    assert torch.equal(px0, px[:1])
    # several blobs at once (their entropy-decoding launches side by side): the same pixels, in order
    xp = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(1, 384, 256, seed=9))).to(dev)
    blobp = model.compress(xp)
    many = model.decompress_many([blob, blobp, blob0])
    assert torch.equal(many[0], px) and torch.equal(many[1], model.decompress(blobp)) and torch.equal(many[2], px0)
    # ... and several batches compressed at once (their launches side by side, two read-backs for the set): the same bytes
    assert model.compress_many([x, xp]) == [blob, blobp] and model.compress_many([xp]) == [blobp] and model.compress_many([]) == []
    # a flipped payload word is refused whether it sits in the hyper-latents' streams or in the latents' (one counter per
    # entropy-decoding launch, summed once)
    import struct as _st
    from shallow_ntc_amd.entropy_coding import Codec as _Codec
    hd = model._get_codec()._parse(blob)
    for word in (hd["pos"] // 2 + 300, hd["pos"] // 2 + hd["zw"] + 5000):
        raw = bytearray(blob)
        raw[2 * word] ^= 0x10
        with pytest.raises(Exception, match="corrupt"):
            model.decompress(bytes(raw))
        with pytest.raises(Exception, match="corrupt"):
            model.decompress_many([blob0, bytes(raw)])
    from shallow_ntc_amd import _capi
    with pytest.raises(_capi.SntcError):
        model.decompress(blob[:-10])
    with pytest.raises(_capi.SntcError):
        model.decompress(b"JUNK" + blob[4:])
    # a header that lies about any dimension is refused before anything is allocated or decoded with it
    import struct
    from shallow_ntc_amd.entropy_coding import Codec
    head = list(struct.unpack_from(Codec.HEAD, blob, 4))       # ver n H W C Cz hz wz h w sz sy lz ly
    assert head[1:10] == [2, 512, 768, 320, 320, 8, 12, 32, 48]
    hsize = struct.calcsize(Codec.HEAD)
    for field, value in ((1, 60000), (2, 100000), (4, 640), (5, 64), (6, 9), (8, 64), (9, 4800), (10, 7), (12, 3)):
        bad = list(head)
        bad[field] = value
        forged = blob[:4] + struct.pack(Codec.HEAD, *bad) + blob[4 + hsize:]
        with pytest.raises(_capi.SntcError, match="header"):
            model.decompress(forged)


def test_gpu_words_equal_the_golden_stream(dev):
    """The HIP encoder emits exactly the committed words of tests/golden/bitstream.npz and decodes them back."""
    from pathlib import Path
    from shallow_ntc_amd import entropy_coding as ec
    g = np.load(Path(__file__).parent / "golden" / "bitstream.npz")
    dt = ec.DeviceTables(ec.normal_tables(), dev)
    vals = torch.from_numpy(g["values"]).to(dev).view(2, -1, 1)
    tids = torch.from_numpy(g["table_ids"]).to(dev).view(2, -1, 1)
    for segs in (1, 3):
        lanes = int(g[f"lanes_s{segs}"])
        payload, lens = ec.rans_encode(vals, tids, dt, segs, lanes)
        assert lens.tolist() == g[f"lens_s{segs}"].tolist()
        np.testing.assert_array_equal(payload.cpu().numpy().view(np.uint16), g[f"words_s{segs}"])
        gold = torch.from_numpy(g[f"words_s{segs}"].view(np.int16)).to(dev)
        back = ec.rans_decode(gold, g[f"lens_s{segs}"], tids, tuple(vals.shape), dt, segs, lanes)
        assert torch.equal(back, vals)


@pytest.mark.gpu
def test_a_flagged_stream_k_launch_raises_in_every_codec_path(dev):
    """A stream-K hand-off that timed out leaves INVALID results and a sticky device flag (include/sntc.h, Stream-K health).
    compress / decompress / encode / decode must raise on it -- never hand out a wrong file or wrong pixels -- and the
    library then runs the static schedule, with which the same calls succeed and agree with the un-flagged results."""
    from shallow_ntc_amd import _capi, ops
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.02))
    x = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(1, 256, 256, seed=3))).to(dev)
    blob = model.compress(x)
    px = model.decompress(blob)
    z_hat, sym, _, _ = model.encode(x)
    inject = lambda: _capi.call("sntc_conv_status_inject", 1, ops._stream())
    try:
        for call in (lambda: model.compress(x), lambda: model.decompress(blob), lambda: model.encode(x),
                     lambda: model.decode(z_hat, sym, (256, 256))):
            inject()
            with pytest.raises(_capi.SntcError, match="stream-K"):
                call()
            # the flag was read and cleared in one atomic; the schedule is static now and the call goes through
            flags = __import__("ctypes").c_int(-1)
            _capi.call("sntc_conv_status", __import__("ctypes").byref(flags), ops._stream())
            assert flags.value == 0
        assert model.compress(x) == blob
        assert torch.equal(model.decompress(blob), px)
        # a deferred check (check=False) leaves the flag for the caller's own synchronisation point
        inject()
        model.decode(z_hat, sym, (256, 256), check=False)
        with pytest.raises(_capi.SntcError, match="stream-K"):
            ops.check_conv_status()
    finally:
        ops.set_stream_k(True, force=True)      # the injected time-outs latched stream-K off; the device is this test's own


def test_many_blobs_of_mixed_sizes_in_any_order(dev):
    """decompress_many pipelines its blobs largest first on the library's pooled streams (more blobs than the pool's three streams:
    the pool grows, queues are shared): whatever the order they are handed in, every blob's pixels are those of its own
    ``decompress``; repeated calls included.  compress_many returns the same bytes as one ``compress`` per batch."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    from shallow_ntc_amd import ops
    was = ops.stream_k_enabled()
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.01))
    shapes = [(3, 256, 384), (1, 384, 256), (2, 128, 128), (1, 512, 768), (2, 200, 120)]
    xs = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=11 + i))).to(dev) for i, (n, h, w) in enumerate(shapes)]
    blobs = [model.compress(x) for x in xs]
    assert model.compress_many(xs) == blobs
    want = [model.decompress(b) for b in blobs]
    rng = np.random.default_rng(0)
    for _ in range(3):
        order = rng.permutation(len(blobs))
        got = model.decompress_many([blobs[i] for i in order])
        for i, px in zip(order, got):
            assert torch.equal(px, want[i]), f"blob {i} in order {list(order)}"
    assert ops.stream_k_enabled() == was                   # the static-schedule blocks left the process-wide switch as they found it

"""Split precision on pre-split operands (csrc/bf3_gemm.hip, Model(precision="bf16x3")) on the GPU against the float64 oracle
(-m gpu): the S3 format, the 256-wide direct-to-LDS kernel over the layer shapes of the decoders (phase-grouped transposed
convolutions, forward convolutions, epilogues), schedule / tile / batch invariance, and the model-level bars of BASELINE.json
(|d bpp| <= 1e-4, |d PSNR| <= 1e-3 dB against the oracle; compress -> decompress round trip in the same arithmetic)."""
import numpy as np
import pytest
import torch

from oracle import model_np
from oracle import ops_np as O

pytestmark = pytest.mark.gpu


def dev_t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def rel_err(got, ref):
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def bf16_round(a):
    """float32 -> nearest-even bfloat16, as float32 (NumPy restatement of the device conversion)."""
    u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def test_split3_format(dev):
    """Format S3: [n, h, w, C / 16, 3, 16] bfloat16 = (hi, mid, lo) with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid):
    bit-exact against the NumPy restatement, and hi + mid + lo reproduces x to 2^-24 relative."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((2, 5, 7, 48)) * np.exp(rng.uniform(-6, 6, size=(2, 5, 7, 48)))).astype(np.float32)
    x[0, 0, 0, :4] = [0.0, -0.0, 1.0, -3.5]
    s3 = ops.split3(dev_t(x, dev))
    assert tuple(s3.shape) == (2, 5, 7, 3, 3, 16) and s3.dtype == torch.bfloat16
    got = s3.float().cpu().numpy()
    xs = x.reshape(2, 5, 7, 3, 16)
    hi = bf16_round(xs)
    mid = bf16_round(xs - hi)
    lo = bf16_round(xs - hi - mid)
    np.testing.assert_array_equal(got[..., 0, :], hi)
    np.testing.assert_array_equal(got[..., 1, :], mid)
    np.testing.assert_array_equal(got[..., 2, :], lo)
    rec = got.astype(np.float64).sum(axis=-2)
    assert np.abs(rec - xs).max() <= 2.0 ** -23 * np.abs(xs).max() and np.all(np.abs(rec - xs) <= 2.0 ** -24 * np.abs(xs) + 1e-45)
    # the decoder's dequantisation fused with the split: symbols + mu
    sym = rng.integers(-9, 9, size=(2, 5, 7, 48)).astype(np.int32)
    hyper = rng.standard_normal((2, 5, 7, 96)).astype(np.float32)
    y3, yf = ops.dequant_split3(torch.from_numpy(sym).to(dev), dev_t(hyper, dev), want_float=True)
    want = sym.astype(np.float32) + hyper[..., :48]
    np.testing.assert_array_equal(yf.cpu().numpy(), want)
    assert torch.equal(y3, ops.split3(dev_t(want, dev)))
    with pytest.raises(ValueError):
        ops.split3(dev_t(x[..., :40], dev))


CASES = [  # kind, k, s, cin, cout, n, h, w, act, epilogue
    ("convT", 3, 1, 64, 96, 3, 20, 24, None, False),         # hyper-synthesis 3 shape family (stride-1 transpose)
    ("convT", 5, 2, 32, 48, 2, 17, 19, "relu", False),       # four phase groups of different K
    ("convT", 13, 8, 32, 24, 2, 18, 16, None, False),        # two-layer synthesis: N = 600 / 360 / 360 / 216 columns, Cout = 24
    ("convT", 18, 16, 32, 4, 1, 17, 16, None, False),        # JPEG-like geometry (Cout padded to 4 for the 16-B epilogue)
    ("conv", 3, 1, 48, 64, 2, 18, 23, "relu", True),         # forward convolution with the ResidualBlock skip
    ("conv", 5, 2, 64, 64, 2, 37, 41, "leaky_relu", False),  # strided, ragged sizes
    ("conv", 1, 1, 96, 192, 2, 16, 17, None, True),
    ("sigup", 5, 2, 32, 32, 1, 16, 17, None, False),         # tfc.SignalConv2D geometry
]


@pytest.mark.parametrize("kind,k,s,cin,cout,n,h,w,act,epi", CASES)
def test_presplit_kernel_against_float64(kind, k, s, cin, cout, n, h, w, act, epi, dev):
    """Pre-split plan == float64 oracle to the fp32 path's own accuracy, for both tile shapes, static and stream-K schedules,
    patch and per-tap staging (bit-identical to each other: every output is the same chain of MFMA terms), and image-alone ==
    image-in-batch."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(k * 100 + cin + cout)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    up = kind in ("convT", "sigup")
    wk = (rng.standard_normal((k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    fn = dict(conv=O.conv2d, convT=O.conv2d_transpose, sigup=O.signal_conv_up)[kind]
    ref = fn(x, wk, b, s)
    if act == "relu":
        ref = O.relu(ref)
    elif act == "leaky_relu":
        ref = O.leaky_relu(ref)
    res = rng.standard_normal(ref.shape).astype(np.float32) if epi else None
    if epi:
        ref = ref + res
    e = capi.EPI_ADD if epi else capi.EPI_STORE
    p32 = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s, act, capi.PRO_NONE, e)
    ps = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s, act, capi.PRO_NONE, e, bf16x3="presplit")
    xd, rd = dev_t(x, dev), (dev_t(res, dev) if epi else None)
    xs = ops.split3(xd)
    e32 = rel_err(p32(xd, res=rd).cpu().numpy(), ref)
    outs = []
    for variant in (11, 12, 13):
        ps.set_tile(variant)
        for sk, halo in ((True, True), (False, True), (True, False), (False, False)):
            ps.set_stream_k(sk, force=sk, halo=halo)      # halo: one activation patch per channel slab where the geometry allows
            y = ps(xs, res=rd)
            outs.append(y)
            e3 = rel_err(y.cpu().numpy(), ref)
            assert e3 < 5e-6 and e3 < 4 * e32 + 2e-7, (variant, sk, halo, e3, e32)
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    ps.set_tile(0)
    ps.set_stream_k(True)
    alone = ps(ops.split3(xd[:1].contiguous()), res=None if rd is None else rd[:1].contiguous())
    assert torch.equal(alone, outs[0][:1])
    with pytest.raises(ValueError):
        ps(xd)                                                      # fp32 input to a pre-split plan: refused, not reinterpreted


def test_presplit_plan_limits(dev):
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(1)
    with pytest.raises(capi.SntcError):                               # Cin % 16 != 0
        ops.ConvPlan("conv", dev_t(rng.standard_normal((5, 5, 3, 16)), dev), None, 2, bf16x3="presplit")
    with pytest.raises(capi.SntcError):                               # Cout % 4 != 0
        ops.ConvPlan("convT", dev_t(rng.standard_normal((5, 5, 3, 32)), dev), None, 2, bf16x3="presplit")
    with pytest.raises(capi.SntcError):                               # GDN epilogues stay on the fp32 path
        ops.ConvPlan("conv", dev_t(rng.standard_normal((1, 1, 32, 32)), dev), None, 1, None, capi.PRO_NONE, capi.EPI_RES_DIV,
                     bf16x3="presplit")


def test_stream_k_chain_at_full_width(dev):
    """The real 480 -> 640 hyper-synthesis layer at Kodak batch size: stream-K (256 workers, every tile cut between two of
    them) == one workgroup per tile, patch staging == per-tap staging, bit for bit, for both tiles; and the sticky status word
    stays clear."""
    from shallow_ntc_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn((6, 32, 48, 480), device=dev, generator=g)
    wk = torch.randn((3, 3, 640, 480), device=dev, generator=g) * 0.02
    b = torch.randn((640,), device=dev, generator=g)
    ps = ops.ConvPlan("convT", wk, b, 1, "relu", bf16x3="presplit")
    xs = ops.split3(x)
    ref = ops.ConvPlan("convT", wk, b, 1, "relu")(x)
    outs = []
    for variant in (11, 12, 13):
        ps.set_tile(variant)
        for sk, halo in ((True, True), (False, True), (True, False), (False, False)):
            ps.set_stream_k(sk, force=sk, halo=halo)
            outs.append(ps(xs))
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    assert float((outs[0] - ref).abs().max() / ref.abs().max()) < 5e-6
    torch.cuda.synchronize()
    ops.check_conv_status()


TC = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 64)),
          synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                         activation_type="igdn", res_type="conv"))


def _models(dev, tc, scale_y=1.0):
    from shallow_ntc_amd.mshyper.models import Model
    m32 = Model(rd_lambda=0.02, transform_config=tc, device=dev, quality_metrics=False)
    w = dict(m32.get_weights())
    rng = np.random.default_rng(5)
    for k in list(w):
        if k.endswith("/bias"):
            w[k] = (0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
    c2 = w["hyper_synthesis/layer_2/bias"].shape[0] // 2
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[c2:] = rng.uniform(-1, 2.5, size=c2)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    m32.set_weights(w)
    from shallow_ntc_amd.common import data_lib
    probe = data_lib.normalize_image(data_lib.synthetic_images(1, 128, 128, seed=99))
    gain = np.float32(scale_y / float(m32.infer_latent_rvs(probe).uq[1].loc.std()))       # latents with std scale_y: a live rate
    w["analysis/conv3/kernel"] = (w["analysis/conv3/kernel"] * gain).astype(np.float32)
    w["analysis/conv3/bias"] = (w["analysis/conv3/bias"] * gain).astype(np.float32)
    m32.set_weights(w)
    m3 = Model(rd_lambda=0.02, transform_config=tc, device=dev, quality_metrics=False, precision="bf16x3")
    m3.set_weights(w)
    return m32, m3, w


@pytest.mark.parametrize("synth", [None, dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16)], ids=["two_layer_res", "jpeg_like"])
def test_model_in_bf16x3_meets_the_baseline_bars(synth, dev, monkeypatch):
    """image -> (bpp, PSNR) with the decoder-side transforms in split precision, against the float64 oracle: symbols differ
    only at the oracle's own near-ties, and at the GPU's integers |d bpp| <= 1e-4, |d PSNR| <= 1e-3 dB (the bars of
    tests/test_hip_model.py); compress -> decompress reproduces decode(encode(x)) bit for bit, and a decoder of the other
    arithmetic refuses the stream."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd.common import _graph, data_lib
    tc = dict(TC) if synth is None else dict(TC, synthesis=synth)
    # this reduced-width model offers the pre-split kernel two tiles per image where the real ones offer 18 - 78: lift the
    # occupancy rule (common/_graph.py::S3_MIN_TILES, a speed rule) so that the arithmetic under test actually runs
    monkeypatch.setattr(_graph, "S3_MIN_TILES", 1)
    m32, m3, w = _models(dev, tc, scale_y=1.5)
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 256, 320, seed=7))          # 16 x 20 latents: >= 256 rows per image
    # two-layer synthesis: Cout = 24 runs pre-split; the JPEG-like layer (Cout = 3: no 16-B epilogue) stays on the fp32 kernel,
    # and so do the hyper-synthesis layers whose maps are smaller than one 256-row strip per image (a per-image rule, never a
    # per-batch one: the decoder must take the same arithmetic as the encoder whatever its batching)
    assert m3._synthesis.takes_s3(16, 20) == (synth is None) and not m32._synthesis.takes_s3(16, 20)
    hs = [l.plan for l in m3._hyper_synthesis._graph.layers]
    assert [p.takes_s3(h, w) for p, (h, w) in zip(hs, [(4, 5), (8, 10), (16, 20)])] == [False, False, True]
    lat = m3.infer_latent_rvs(x)
    r = m3._rate_and_reconstruction(lat, want_symbols=True)
    sym = r["symbols"].cpu().numpy()
    _, metrics = m3.frame_loss_given_latent_rvs(x, lat, training=False)
    got = metrics.scalars_float
    ref_model = model_np.Model(tc, rd_lambda=0.02)
    ref = ref_model.end_to_end(w, x)
    flip = sym != ref["symbols_y"]
    at = ref_model.frame_loss(w, x, ref_model.infer_latents(w, x), force_symbols=sym)
    assert int(flip.sum()) <= 4 and (at["tie_distance"][flip] < 1e-3).all(), int(flip.sum())
    assert abs(got["bpp"] - at["bpp"]) <= 1e-4, (got["bpp"], at["bpp"])
    assert abs(got["psnr"] - at["psnr"]) <= 1e-3, (got["psnr"], at["psnr"])
    assert got["bpp"] > 0.3                                                           # a live operating point
    # codec in bf16x3: encoder and decoder derive the same table ids -> bit-exact round trip; decode batches of other sizes
    blob = m3.compress(x)
    px = m3.decompress(blob)
    z_hat, symbols, _, _ = m3.encode(x)
    assert torch.equal(px, m3.decode(z_hat, symbols, (256, 320)))
    blob1 = m3.compress(x[:1])
    one = m3.decompress(blob1)
    assert torch.equal(one, px[:1])                                                  # batch-invariant arithmetic
    many = m3.decompress_many([blob1, blob])                                         # decoding launches side by side: same pixels
    assert torch.equal(many[0], one) and torch.equal(many[1], px)
    with pytest.raises(capi.SntcError, match="bf16x3"):
        m32.decompress(blob)
    # and it is close to, but not the same arithmetic as, the fp32 model
    p32 = m32.decode(*m32.encode(x)[:2], (256, 320))
    d = (p32.to(torch.int16) - px.to(torch.int16)).abs()
    assert int(d.max()) <= 1


def test_split_precision_analysis_does_not_depend_on_the_batch(dev):
    """Under precision="bf16x3" the arithmetic a layer takes is a function of the layer and of ONE image's geometry (DualPlan;
    the ResidualBlocks run the exact fp32 block or layers, which are bit-identical to each other): an image's latents are
    the same bits alone and inside a batch of four, and so are its decoded pixels."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, precision="bf16x3", **configs.two_layer_syn(rd_lambda=0.02))
    x = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(4, 256, 384, seed=11))).to(dev)
    z4, s4, _, _ = model.encode(x)
    z1, s1, _, _ = model.encode(x[2:3].contiguous())
    assert torch.equal(z4[2:3], z1) and torch.equal(s4[2:3], s1)
    p4 = model.decode(z4, s4, (256, 384))
    p1 = model.decode(z1, s1, (256, 384))
    assert torch.equal(p4[2:3], p1)


@pytest.mark.parametrize("n,h,w", [(1, 8, 32), (2, 13, 45), (1, 64, 96), (3, 37, 100), (2, 40, 33)])
def test_whole_residual_block_in_split_precision(n, h, w, dev):
    """sntc_resblock_forward with precision 1 (csrc/rb_fused_bf3.hip; reference common/elic.py:41-68): the whole c = 192 block in
    bf16 x 3 on pre-split weights, pixels split in registers -- fp32-level accuracy against the float64 oracle (within 4x the exact
    fp32 block's own error), the same bits for any number of persistent workgroups and for an image alone or inside a batch."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(77 * n + h + w)
    c = 192
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    mk = lambda scale, *shape: (rng.standard_normal(shape) * scale).astype(np.float32)
    w0, b0 = mk(0.08, 1, 1, c, c // 2), mk(1.0, c // 2)
    w1, b1 = mk(0.05, 3, 3, c // 2, c // 2), mk(1.0, c // 2)
    w2, b2 = mk(0.1, 1, 1, c // 2, c), mk(1.0, c)
    t = np.maximum(O.conv2d(x.astype(np.float64), w0, b0, 1), 0.0)
    t = np.maximum(O.conv2d(t, w1, b1, 1), 0.0)
    ref = x + O.conv2d(t, w2, b2, 1)
    args = [dev_t(a, dev) for a in (w0, b0, w1, b1, w2, b2)]
    xd = dev_t(x, dev)
    exact = ops.ResBlockPlan(*args)
    split = ops.ResBlockPlan(*args, precision="bf16x3")
    e32 = rel_err(exact(xd).cpu().numpy(), ref)
    y = split(xd)
    e3 = rel_err(y.cpu().numpy(), ref)
    assert e3 < 5e-6 and e3 < 4 * e32 + 2e-7, (e3, e32)
    for wg in (1, 3, 200):
        split.set_workgroups(wg)
        assert torch.equal(split(xd), y), wg
    split.set_workgroups(0)
    assert torch.equal(split(xd[:1].contiguous()), y[:1])

"""GPU parity of every C-ABI op against the float64 oracle on seeded inputs (-m gpu)."""
import zlib

import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def dev_t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


# float32 accumulation against float64: K up to ~8000 products per output
TOL = 2e-5

CONV_CASES = [  # kind, k, s, cin, cout, n, h, w, act
    ("conv", 1, 1, 32, 64, 2, 9, 7, "relu"),
    ("conv", 3, 1, 32, 32, 1, 16, 12, "relu"),
    ("conv", 3, 1, 96, 96, 1, 20, 24, None),
    ("conv", 5, 2, 64, 96, 2, 16, 20, None),
    ("conv", 5, 2, 32, 160, 1, 17, 13, "leaky_relu"),     # odd sizes: SAME pad changes
    ("conv", 5, 2, 3, 32, 2, 32, 24, None),               # first layer, Cin = 3 (scalar-gather path)
    ("conv", 1, 1, 192, 96, 1, 12, 16, "sigmoid"),
    ("conv", 5, 2, 320, 320, 1, 8, 12, "relu"),
    ("sigdown", 5, 2, 32, 64, 1, 16, 16, None),
    ("sigdown", 9, 4, 3, 32, 1, 32, 32, None),
    ("sigdown", 3, 1, 32, 32, 1, 7, 9, "relu"),
    ("convT", 5, 2, 32, 64, 2, 6, 5, "relu"),
    ("convT", 5, 2, 320, 480, 1, 4, 6, "relu"),
    ("convT", 3, 1, 96, 128, 1, 8, 12, None),
    ("convT", 13, 8, 64, 24, 1, 4, 6, None),
    ("convT", 18, 16, 64, 3, 2, 3, 4, None),
    ("convT", 16, 16, 32, 3, 1, 2, 3, None),
    ("convT", 6, 4, 32, 64, 1, 3, 3, None),
    ("sigup", 5, 2, 32, 32, 1, 5, 6, None),
    ("sigup", 9, 4, 32, 3, 1, 4, 4, None),
    ("sigup", 3, 1, 32, 64, 1, 6, 5, "relu"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(map(str, c)))
def test_conv_family(case, dev):
    from shallow_ntc_amd import ops
    kind, k, s, cin, cout, n, h, w, act = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wshape = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
    wk = (rng.standard_normal(wshape) / np.sqrt(k * k * cin / 4)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    fn = {"conv": O.conv2d, "convT": O.conv2d_transpose, "sigdown": O.signal_conv_down, "sigup": O.signal_conv_up}[kind]
    ref = O.ACTIVATIONS[act](fn(x, wk, b, s))
    plan = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s, act)
    got = plan(dev_t(x, dev)).cpu().numpy()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL
    assert plan.flops(n, h, w) == 2 * n * k * k * cin * cout * (h * w if kind in ("convT", "sigup") else ref.shape[1] * ref.shape[2])


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_every_tile_variant(variant, dev):
    """Each gather-GEMM instantiation gives the same answer (tile selection is a speed choice only)."""
    from shallow_ntc_amd import _capi, ops
    rng = np.random.default_rng(variant)
    x = rng.standard_normal((2, 11, 13, 64)).astype(np.float32)
    wk = (rng.standard_normal((5, 5, 200, 64)) * 0.05).astype(np.float32)     # convT 5/2: 4 phase groups
    b = rng.standard_normal(200).astype(np.float32)
    ref = O.conv2d_transpose(x, wk, b, 2)
    plan = ops.ConvPlan("convT", dev_t(wk, dev), dev_t(b, dev), 2)
    plan.set_tile(variant)
    assert plan.launch_info(2, 11, 13)[0] == variant
    got = plan(dev_t(x, dev)).cpu().numpy()
    assert rel_err(got, ref) < TOL


@pytest.mark.parametrize("n,h,w,stream_k", [(1, 9, 13, True), (2, 40, 56, True), (18, 64, 96, True), (18, 64, 96, False),
                                            (3, 37, 29, False)])
def test_fused_residual_block_tail_is_bit_identical(n, h, w, stream_k, dev):
    """conv3x3(96 -> 96, relu) -> conv1x1(96 -> 192) + skip in ONE launch (sntc_conv_forward_fused; reference
    common/elic.py:57-68) gives exactly the bits of the two launches -- ragged row counts, stream-K and static schedules,
    with and without the skip -- and matches the float64 oracle; weights updated in place are picked up."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(n * 100 + h)
    x = dev_t(rng.standard_normal((n, h, w, 96)).astype(np.float32), dev)
    w1 = (rng.standard_normal((3, 3, 96, 96)) * 0.05).astype(np.float32)
    b1 = rng.standard_normal(96).astype(np.float32)
    w2 = (rng.standard_normal((1, 1, 96, 192)) * 0.1).astype(np.float32)
    b2 = rng.standard_normal(192).astype(np.float32)
    res = dev_t(rng.standard_normal((n, h, w, 192)).astype(np.float32), dev)
    first = ops.ConvPlan("conv", dev_t(w1, dev), dev_t(b1, dev), 1, "relu")
    for epi in (capi.EPI_ADD, capi.EPI_STORE):
        second = ops.ConvPlan("conv", dev_t(w2, dev), dev_t(b2, dev), 1, None, capi.PRO_NONE, epi)
        assert first.fusable_with(second)
        first.set_stream_k(stream_k)
        r = res if epi == capi.EPI_ADD else None
        two = second(first(x), res=r)
        one = first.fused(second, x, res=r)
        assert torch.equal(one, two)
    if n <= 2:
        ref = O.conv2d(np.maximum(O.conv2d(x.cpu().numpy(), w1, b1, 1), 0.0), w2, b2, 1)
        assert rel_err(one.cpu().numpy(), ref) < TOL
    # a weight update of the 1x1 plan reaches the fused path (its fragment-order copy is rebuilt)
    w2b = dev_t((w2 * 0.5).astype(np.float32), dev)
    second.update(w2b, dev_t(b2, dev))
    assert torch.equal(first.fused(second, x), second(first(x)))
    # pairs that do not qualify are refused, not mis-computed
    other = ops.ConvPlan("conv", dev_t(w2, dev), dev_t(b2, dev), 1, "relu")
    assert not first.fusable_with(other)
    with pytest.raises(capi.SntcError):
        first.fused(other, x)


@pytest.mark.parametrize("n,h,w", [(1, 8, 32), (2, 13, 45), (1, 64, 96), (3, 37, 100), (1, 5, 7), (2, 40, 33)])
def test_whole_residual_block_in_one_launch_is_bit_identical(n, h, w, dev):
    """sntc_resblock_forward (reference common/elic.py:41-68: x + conv1x1(relu(conv3x3(relu(conv1x1(x)))))) on an 8 x 32 pixel
    tile with its halo patch in LDS gives exactly the bits of the three gather-GEMM launches -- ragged sizes, image borders
    inside and between tiles, any number of persistent workgroups, with and without biases -- and matches the float64 oracle;
    a weight update is picked up; unsupported widths and aliased buffers are refused."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(1000 * n + 10 * h + w)
    c = 192
    x = dev_t(rng.standard_normal((n, h, w, c)).astype(np.float32), dev)
    mk = lambda scale, *shape: (rng.standard_normal(shape) * scale).astype(np.float32)
    w0, b0 = mk(0.08, 1, 1, c, c // 2), mk(1.0, c // 2)
    w1, b1 = mk(0.05, 3, 3, c // 2, c // 2), mk(1.0, c // 2)
    w2, b2 = mk(0.1, 1, 1, c // 2, c), mk(1.0, c)
    for with_bias in (True, False):
        bs = [dev_t(b, dev) if with_bias else None for b in (b0, b1, b2)]
        ws = [dev_t(a, dev) for a in (w0, w1, w2)]
        la = ops.ConvPlan("conv", ws[0], bs[0], 1, "relu")
        lb = ops.ConvPlan("conv", ws[1], bs[1], 1, "relu")
        lc = ops.ConvPlan("conv", ws[2], bs[2], 1, None, capi.PRO_NONE, capi.EPI_ADD)
        three = lc(lb(la(x)), res=x)
        block = ops.ResBlockPlan(ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])
        one = block(x)
        assert torch.equal(one, three), float((one - three).abs().max())
        for wg in (1, 3, 8, 200):
            block.set_workgroups(wg)
            assert torch.equal(block(x), three), wg
        block.set_workgroups(0)
    if n * h * w <= 4096:
        t = np.maximum(O.conv2d(x.cpu().numpy().astype(np.float64), w0, None, 1), 0.0)
        t = np.maximum(O.conv2d(t, w1, None, 1), 0.0)
        ref = x.cpu().numpy() + O.conv2d(t, w2, None, 1)
        assert rel_err(one.cpu().numpy(), ref) < TOL
    block.update(dev_t(w0 * 0.5, dev), None, ws[1], None, ws[2], None)
    la.update(dev_t(w0 * 0.5, dev))
    assert torch.equal(block(x), lc(lb(la(x)), res=x))
    assert not ops.ResBlockPlan.supported(64)
    with pytest.raises(capi.SntcError):
        capi.call("sntc_resblock_forward", block._h, ops._ptr(x), n, h, w, ops._ptr(x), ops._stream())


def test_plan_group_update_equals_per_plan_update(dev):
    """sntc_plan_group_update re-packs every plan of the group in one launch exactly as sntc_conv_plan_update does one by
    one: forward convolution, phase-grouped transpose (four weight groups), input-gradient plan on the swapped kernel,
    dword-gather first layer, and the fused-tail fragment copy of a 1x1 96 -> 192 plan."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(21)
    mk = lambda *shape: dev_t((rng.standard_normal(shape) * 0.1).astype(np.float32), dev)
    specs = [("conv", (3, 3, 96, 96), 96, 1, "relu", capi.EPI_STORE, False), ("convT", (5, 5, 48, 32), 48, 2, None, capi.EPI_STORE, False),
             ("convT", (3, 3, 96, 96), None, 1, None, capi.EPI_STORE, True), ("conv", (5, 5, 3, 64), 64, 2, None, capi.EPI_STORE, False),
             ("conv", (1, 1, 96, 192), 192, 1, None, capi.EPI_ADD, False)]
    entries, fresh_args = [], []
    for kind, wshape, nb, s, act, epi, swapped in specs:
        w, b = mk(*wshape), (mk(nb) if nb else None)
        entries.append((ops.ConvPlan(kind, w, b, s, act, capi.PRO_NONE, epi, kernel_io_swapped=swapped), w, b))
        fresh_args.append((kind, s, act, epi, swapped))
    group = ops.PlanGroup(entries)
    for _p, w, b in entries:                       # new values in the SAME arrays, as the optimizer leaves them
        w.copy_(mk(*w.shape))
        if b is not None:
            b.copy_(mk(*b.shape))
    group.update()
    x96, x32, x3 = mk(2, 9, 11, 96), mk(2, 9, 11, 32), mk(2, 20, 22, 3)
    res = mk(2, 9, 11, 192)
    inputs = [x96, x32, x96, x3, x96]
    for (plan, w, b), (kind, s, act, epi, swapped), x in zip(entries, fresh_args, inputs):
        fresh = ops.ConvPlan(kind, w, b, s, act, capi.PRO_NONE, epi, kernel_io_swapped=swapped)
        r = res if epi == capi.EPI_ADD else None
        assert torch.equal(plan(x, res=r), fresh(x, res=r)), kind
    first, second = entries[0][0], entries[4][0]
    assert torch.equal(first.fused(second, x96, res=res), second(first(x96), res=res))


def test_epilogues(dev):
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 7, 32)).astype(np.float32)
    wk = (rng.standard_normal((1, 1, 32, 64)) * 0.2).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    res = rng.standard_normal((2, 6, 7, 64)).astype(np.float32)
    aux = rng.standard_normal((2, 6, 7, 64)).astype(np.float32)
    conv = O.conv2d(x, wk, b, 1)
    add = ops.ConvPlan("conv", dev_t(wk, dev), dev_t(b, dev), 1, None, capi.PRO_NONE, capi.EPI_ADD)
    assert rel_err(add(dev_t(x, dev), res=dev_t(res, dev)).cpu().numpy(), conv + res) < TOL
    gate = ops.ConvPlan("conv", dev_t(wk, dev), dev_t(b, dev), 1, "sigmoid", capi.PRO_NONE, capi.EPI_GATE)
    got = gate(dev_t(x, dev), res=dev_t(res, dev), aux=dev_t(aux, dev)).cpu().numpy()
    assert rel_err(got, res + aux * O.sigmoid(conv)) < TOL
    with pytest.raises(capi.SntcError):
        add(dev_t(x, dev))          # epilogue operand missing -> loud failure, not a silent store


@pytest.mark.parametrize("c,inverse,alpha,eps", [(12, True, 1, 1.0), (12, False, 1, 1.0), (24, True, 1, 1.0),
                                                 (48, False, 2, 0.5), (64, False, 1, 1.0), (192, True, 1, 1.0),
                                                 (96, False, 2, 0.5), (256, False, 1, 1.0)])
def test_gdn(c, inverse, alpha, eps, dev):
    from shallow_ntc_amd.common._graph import GDN
    rng = np.random.default_rng(c)
    x = rng.standard_normal((2, 9, 10, c)).astype(np.float32)
    beta = (1.0 + rng.random(c)).astype(np.float32)
    gamma = (0.1 * np.eye(c) + 0.02 * rng.random((c, c))).astype(np.float32)
    node = GDN("g", inverse, alpha, eps)
    node.build({"g/beta": dev_t(beta, dev), "g/gamma": dev_t(gamma, dev)}, c)
    got = node(dev_t(x, dev)).cpu().numpy()
    assert rel_err(got, O.gdn(x, beta, gamma, inverse, alpha, eps)) < TOL


@pytest.mark.parametrize("ch,has_res,act", [(12, True, "igdn"), (24, False, "igdn"), (48, False, "igdn"),
                                            (12, True, None), (24, True, "relu")])
def test_two_layer_tail(ch, has_res, act, dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(ch)
    n, hh, wh = 2, 21, 19
    t = rng.standard_normal((n, hh, wh, ch * (2 if has_res else 1))).astype(np.float32)
    beta = (1.0 + rng.random(ch)).astype(np.float32)
    gamma = (0.1 * np.eye(ch) + 0.02 * rng.random((ch, ch))).astype(np.float32)
    w2 = (rng.standard_normal((5, 5, 3, ch)) * 0.1).astype(np.float32)
    b2 = rng.standard_normal(3).astype(np.float32)
    base = t[..., :ch].astype(np.float64)
    if act == "igdn":
        base = O.gdn(base, beta, gamma, inverse=True)
    elif act == "relu":
        base = O.relu(base)
    hsum = base + (t[..., ch:] if has_res else 0.0)
    ref = O.conv2d_transpose(hsum, w2, b2, 2)
    got = ops.two_layer_tail(dev_t(t, dev), ch, has_res, ops.TAIL_ACTS[act], dev_t(beta, dev), dev_t(gamma, dev),
                             dev_t(w2, dev), dev_t(b2, dev)).cpu().numpy()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL


@pytest.mark.parametrize("wd", [50, 52])        # 52: rows of 16-B multiples -> the four-values-per-thread pixel kernel
def test_pad_crop_pixels(wd, dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 256, size=(2, 37, wd, 3)).astype(np.float32) / np.float32(255) - np.float32(0.5))
    xp = ops.pad_reflect(dev_t(x, dev), 64, 64)
    np.testing.assert_array_equal(xp.cpu().numpy(), O.pad_images(x, 64).astype(np.float32))
    np.testing.assert_array_equal(ops.crop(xp, 37, wd).cpu().numpy(), x)
    xh = (x + rng.normal(0, 0.05, size=x.shape)).astype(np.float32)
    xh[0, 0, 0, 0] = 2.0      # saturates to 255
    xh[0, 0, 0, 1] = -3.0     # saturates to 0
    xh[0, 0, 1, 0] = np.float32(100.5) / np.float32(255) - np.float32(0.5)   # lands near a .5 tie
    xh_p = np.pad(xh, ((0, 0), (0, 27), (0, 64 - wd), (0, 0)))
    sse, px = ops.pixels_sse(dev_t(x, dev), dev_t(xh_p, dev), want_pixels=True)
    ref_px = O.floats_to_pixels(xh, training=False)
    ref_x = O.floats_to_pixels(x, training=False)
    np.testing.assert_array_equal(px.cpu().numpy(), ref_px)
    ref_sse = ((ref_x.astype(np.int64) - ref_px.astype(np.int64)) ** 2).reshape(2, -1).sum(1)
    np.testing.assert_array_equal(sse.cpu().numpy(), ref_sse)
    np.testing.assert_array_equal(ops.to_pixels(dev_t(xh_p, dev), 37, wd).cpu().numpy(), ref_px)
    fs = ops.float_sse(dev_t(x, dev), dev_t(xh_p, dev)).cpu().numpy()
    ref_fs = (((x.astype(np.float64) - xh.astype(np.float64)) * 255.0) ** 2).reshape(2, -1).sum(1)
    assert np.abs(fs - ref_fs).max() / ref_fs.max() < 1e-5


def _synthetic_latents(rng, n, h, w, c):
    """SURVEY.md 8d entropy set: y-mu ~ Laplace(0,2), mu ~ N(0,1), raw ~ U(-3, 4.3) (spans both clamps)."""
    mu = rng.standard_normal((n, h, w, c)).astype(np.float32)
    raw = rng.uniform(-3.0, 4.3, size=(n, h, w, c)).astype(np.float32)
    y = (mu + rng.laplace(0, 2.0, size=(n, h, w, c))).astype(np.float32)
    return y, np.concatenate([mu, raw], axis=-1)


def test_entropy_scale_normal(dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(11)
    y, hyper = _synthetic_latents(rng, 3, 8, 6, 64)
    y[0, 0, 0, :4] = hyper[0, 0, 0, :4] + np.array([20.0, -35.0, 0.5, -1.5], np.float32)    # far tails + ties
    hyper[0, 0, 0, 64:68] = np.array([-10.0, -10.0, 5.0, 0.0], np.float32)                   # sigma = 0.11 / 256 clamps
    c = 64
    mu, raw = hyper[..., :c], hyper[..., c:]
    ref_yhat, ref_bits, ref_sym = O.scale_indexed_normal(y, mu, np.exp(raw.astype(np.float64)))
    y_hat, bits, sym = ops.entropy_scale_normal(dev_t(y, dev), dev_t(hyper, dev), want_symbols=True)
    # symbols: bit-exact in the integer domain given identical float32 (y, mu)
    ref_sym32 = np.rint(y - mu).astype(np.int32)
    np.testing.assert_array_equal(sym.cpu().numpy(), ref_sym32)
    np.testing.assert_array_equal(y_hat.cpu().numpy(), ref_sym32.astype(np.float32) + mu)
    got = bits.cpu().numpy()
    assert np.abs(got - ref_bits).max() / np.abs(ref_bits).max() < 2e-5
    # explicit-sample mode
    _, bits2, _ = ops.entropy_scale_normal(y_hat, dev_t(hyper, dev), values_only=True)
    assert np.abs(bits2.cpu().numpy() - ref_bits).max() / np.abs(ref_bits).max() < 2e-5
    # decoder-side dequantisation reproduces y_hat exactly
    np.testing.assert_array_equal(ops.dequant_scale_normal(sym, dev_t(hyper, dev)).cpu().numpy(), y_hat.cpu().numpy())


def test_entropy_known_answers(dev):
    """float64 SciPy values of -log2[Phi((v+.5)/s) - Phi((v-.5)/s)]."""
    from shallow_ntc_amd import ops
    kat = {0.0: [7.908418055e-06, 18.47694976, 378.4317401, 22677.58857],
           10.0: [0.2937470448, 3.441034068, 35.88577211, 1941.606578],
           31.5: [3.735669186, 3.761209505, 3.965532119, 13.95192272],
           63.0: [9.325748982, 9.325759989, 9.325848044, 9.330151732]}
    vs = [0.0, 1.0, -3.0, 20.0]
    for idx, want in kat.items():
        for v, w in zip(vs, want):
            y = np.full((1, 1, 1, 4), v, np.float32)
            raw = np.full((1, 1, 1, 4), np.log(idx) if idx > 0 else -50.0, np.float32)
            hyper = np.concatenate([np.zeros_like(y), raw], -1)
            _, bits, _ = ops.entropy_scale_normal(dev_t(y, dev), dev_t(hyper, dev))
            got = bits.cpu().numpy()[0] / 4
            assert abs(got - w) <= 2e-4 * max(w, 1e-3), (idx, v, got, w)


@pytest.mark.parametrize("num_filters", [(3, 3), (3, 3, 3), (2,), (4, 4, 4, 4)])
def test_entropy_factorized(num_filters, dev):
    from oracle import model_np
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(len(num_filters))
    c = 48
    p = model_np.init_deep_factorized(c, rng, num_filters)
    for k in p:                                   # move off the initial values so tanh factors matter
        p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
    nl = len(num_filters) + 1
    ms = [p[f"prior/matrix_{k}"] for k in range(nl)]
    bs = [p[f"prior/bias_{k}"] for k in range(nl)]
    fs = [p[f"prior/factor_{k}"] for k in range(nl - 1)]
    z = (rng.standard_normal((2, 5, 7, c)) * 3).astype(np.float32)
    z[0, 0, 0, :3] = [40.0, -55.0, 2.5]
    ref_v, ref_bits = O.batched_deep_factorized(z, ms, bs, fs)
    prior = ops.DeepFactorizedPrior(ms, bs, fs)
    z_hat, bits = prior(dev_t(z, dev))
    np.testing.assert_array_equal(z_hat.cpu().numpy(), np.rint(z))
    assert np.abs(bits.cpu().numpy() - ref_bits).max() / np.abs(ref_bits).max() < 2e-5
    _, bits2 = prior(z_hat, values_only=True)
    assert np.abs(bits2.cpu().numpy() - ref_bits).max() / np.abs(ref_bits).max() < 2e-5


@pytest.mark.parametrize("num_filters,c,shape", [((3, 3), 320, (3, 8, 12)), ((3, 3, 3), 320, (2, 5, 7)), ((3, 3), 48, (1, 40, 33)),
                                                 ((3, 3), 260, (2, 1, 1))])
def test_entropy_factorized_channel_per_thread_kernel(num_filters, c, shape, dev):
    """Round 6: priors of widths 1 -> 3 -> ... -> 3 -> 1 (what the reference builds, mshyper/models.py:135) take a kernel in which a
    thread owns a channel (csrc/entropy.hip factorized_fast_kernel): channel counts across the 256-thread block boundary, pixel
    counts that do not fill the last chunk, the explicit-sample mode, far tails and ties -- against the float64 oracle, and the
    same integers as ever."""
    from oracle import model_np
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(c + shape[1])
    p = model_np.init_deep_factorized(c, rng, num_filters)
    for k in p:
        p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
    nl = len(num_filters) + 1
    ms = [p[f"prior/matrix_{k}"] for k in range(nl)]
    bs = [p[f"prior/bias_{k}"] for k in range(nl)]
    fs = [p[f"prior/factor_{k}"] for k in range(nl - 1)]
    z = (rng.standard_normal(shape + (c,)) * 3).astype(np.float32)
    z.reshape(-1)[:6] = [40.0, -55.0, 2.5, -0.5, 1.5, 0.49999997]
    ref_v, ref_bits = O.batched_deep_factorized(z, ms, bs, fs)
    prior = ops.DeepFactorizedPrior(ms, bs, fs)
    z_hat, bits = prior(dev_t(z, dev))
    np.testing.assert_array_equal(z_hat.cpu().numpy(), np.rint(z))
    assert np.abs(bits.cpu().numpy() - ref_bits).max() / np.abs(ref_bits).max() < 2e-6
    _, bits2 = prior(z_hat, values_only=True)
    assert np.abs(bits2.cpu().numpy() - ref_bits).max() / np.abs(ref_bits).max() < 2e-6
    # non-integer samples (the explicit-sample mode of the training / SGA paths' callers)
    zs = (z + rng.uniform(-0.5, 0.5, z.shape)).astype(np.float32)
    want = -O.deep_factorized_logprob(zs.astype(np.float64), ms, bs, fs).sum(axis=(1, 2, 3)) / np.log(2)
    _, bits3 = prior(dev_t(zs, dev), values_only=True)
    assert np.abs(bits3.cpu().numpy() - want).max() / np.abs(want).max() < 2e-6


@pytest.mark.parametrize("n,h,w,c", [(1, 1, 1, 4), (2, 3, 5, 12), (3, 7, 9, 320), (1, 33, 17, 8), (2, 64, 48, 320), (5, 16, 16, 20),
                                     (1, 130, 257, 64)])
def test_entropy_scale_normal_shapes(n, h, w, c, dev):
    """The rewritten scan walks (pixel, channel quad) incrementally by the launch's stride: every element visited exactly once for
    channel counts that do not divide the stride, vectors that do not fill the last step, several images -- symbols exact, bits
    against float64, per image."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(n * 1000 + h * 10 + c)
    y, hyper = _synthetic_latents(rng, n, h, w, c)
    mu, raw = hyper[..., :c], hyper[..., c:]
    _, ref_bits, _ = O.scale_indexed_normal(y, mu, np.exp(raw.astype(np.float64)))
    y_hat, bits, sym = ops.entropy_scale_normal(dev_t(y, dev), dev_t(hyper, dev), want_symbols=True)
    ref_sym = np.rint(y - mu).astype(np.int32)
    np.testing.assert_array_equal(sym.cpu().numpy(), ref_sym)
    np.testing.assert_array_equal(y_hat.cpu().numpy(), ref_sym.astype(np.float32) + mu)
    got = bits.cpu().numpy()
    assert got.shape == (n,) and np.abs(got - ref_bits).max() <= 2e-5 * np.abs(ref_bits).max() + 1e-3
    y2, bits2, none = ops.entropy_scale_normal(dev_t(y, dev), dev_t(hyper, dev))
    assert none is None and torch.equal(y2, y_hat) and torch.allclose(bits2, bits, rtol=1e-12, atol=0)   # double atomics: the order of the block sums is free
    # explicit non-integer samples (the reference formulation kept for this mode)
    ys = (y + rng.uniform(-0.4, 0.4, y.shape)).astype(np.float32)
    sig = O.scale_fn(np.clip(np.exp(raw.astype(np.float64)), 0, 63))
    want = -(O.noisy_normal_logprob(ys.astype(np.float64) - mu, sig)).sum(axis=(1, 2, 3)) / np.log(2)
    _, bits3, _ = ops.entropy_scale_normal(dev_t(ys, dev), dev_t(hyper, dev), values_only=True)
    assert np.abs(bits3.cpu().numpy() - want).max() <= 3e-5 * np.abs(want).max() + 1e-3


@pytest.mark.parametrize("axis", [0, 1])
def test_rgb_first_layer_kernel_signal_conv(axis, dev):
    """The first-layer kernel as tfc.SignalConv2D(corr=True, strides_down=2, "same_zeros") -- MBT2018Analysis' first layer (reference
    common/transforms.py:152-155, BASELINE configs[1]): the centred padding origin, against the hand-worked integers of
    tests/test_oracle_pins.py (exact), the float64 oracle and the generic plan (tolerance: another order of summation), an image
    alone == the image in a batch, and the transform picks it."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common import transforms as TR
    from tests.test_oracle_pins import KERAS_DOWN, SIG_DOWN, SIG_W5, SIG_X6
    cout = 192
    scale = np.arange(1, cout + 1, dtype=np.float32)
    for kind, jc, want in (("sigdown", 2, SIG_DOWN), ("conv", 1, KERAS_DOWN)):
        # the pattern along `axis` in channel 0 of line 2 of the other axis; the kernel's delta at jc on that axis lands it on output line 1
        x = np.zeros((1, 6, 6, 3), np.float32)
        w = np.zeros((5, 5, 3, cout), np.float32)
        for a_, v in enumerate(SIG_X6):
            x[(0, a_, 2, 0) if axis == 0 else (0, 2, a_, 0)] = v
        for j, v in enumerate(SIG_W5):
            w[(j, jc, 0) if axis == 0 else (jc, j, 0)] = v * scale
        y = ops.RgbConvPlan(dev_t(w, dev), None, 2, None, kind)(dev_t(x, dev)).cpu().numpy()[0]          # [3, 3, cout]
        line = y[:, 1] if axis == 0 else y[1, :]
        np.testing.assert_array_equal(line, np.asarray(want, np.float32)[:, None] * scale[None, :], err_msg=kind)
        rest = np.delete(y, 1, axis=1 if axis == 0 else 0)
        assert not rest.any(), kind
    rng = np.random.default_rng(77 + axis)
    for n, h, w_, co, act in ((2, 64, 96, 192, None), (1, 37, 41, 128, "relu"), (3, 16, 18, 256, None)):
        x = rng.standard_normal((n, h, w_, 3)).astype(np.float32)
        wk = (rng.standard_normal((5, 5, 3, co)) * 0.2).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        ref = O.ACTIVATIONS[act](O.signal_conv_down(x, wk, b, 2))
        plan = ops.RgbConvPlan(dev_t(wk, dev), dev_t(b, dev), 2, act, "sigdown")
        gen = ops.ConvPlan("sigdown", dev_t(wk, dev), dev_t(b, dev), 2, act)
        got = plan(dev_t(x, dev))
        assert tuple(got.shape) == ref.shape
        e_new, e_gen = rel_err(got.cpu().numpy(), ref), rel_err(gen(dev_t(x, dev)).cpu().numpy(), ref)
        assert e_new < 2e-6 and e_new < 4 * e_gen + 2e-7, (e_new, e_gen)
        assert torch.equal(plan(dev_t(x[:1], dev)), got[:1])
    t = TR.MBT2018Analysis(192, output_channels=320)
    t(dev_t(rng.standard_normal((1, 64, 64, 3)).astype(np.float32) * 0.3, dev))
    assert isinstance(t._graph.layers[0].plan, ops.RgbConvPlan) and t._graph.layers[0].plan.kind == "sigdown"


@pytest.mark.parametrize("kind", ["convT", "sigup"])
def test_small_output_transposed_conv_kernel(kind, dev):
    """csrc/up_small.hip: the syntheses' last layer (5 x 5 / 2 transposed convolution to the 3 image channels; reference
    common/transforms.py:172-175 MBT2018Synthesis -- tfc.SignalConv2D(strides_up=2) --, :195-206 CNNSynthesis -- Keras
    Conv2DTranspose) on the vector ALU: against the float64 oracle and the gather-GEMM plan it replaces (another order of
    summation: tolerance), odd sizes, tiles that hang over the image, channel counts from one slab to 320, with and without bias;
    the hand-worked origin integers of tests/test_oracle_pins.py exactly; an image alone == the image in a batch; the transform
    picks it."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common import transforms as TR
    from tests.test_oracle_pins import KERAS_UP, SIG_UP, SIG_W5, SIG_X6
    fn = O.conv2d_transpose if kind == "convT" else O.signal_conv_up
    assert ops.UpSmallPlan.supported(kind, 5, 2, 192, 3) and not ops.UpSmallPlan.supported(kind, 5, 2, 192, 12)
    assert not ops.UpSmallPlan.supported(kind, 3, 1, 192, 3) and not ops.UpSmallPlan.supported("conv", 5, 2, 192, 3)
    assert not ops.UpSmallPlan.supported(kind, 5, 2, 24, 3)
    rng = np.random.default_rng(5 + len(kind))
    for n, h, w, cin, bias in ((2, 16, 20, 192, True), (1, 5, 7, 32, True), (1, 33, 17, 64, False), (3, 8, 8, 320, True), (1, 1, 1, 16, True),
                               (2, 40, 31, 192, True)):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wshape = (5, 5, 3, cin) if kind == "convT" else (5, 5, cin, 3)
        wk = (rng.standard_normal(wshape) / np.sqrt(25 * cin / 4)).astype(np.float32)
        b = rng.standard_normal(3).astype(np.float32) if bias else None
        ref = fn(x, wk, b, 2)
        plan = ops.UpSmallPlan(kind, dev_t(wk, dev), None if b is None else dev_t(b, dev), 2)
        gen = ops.ConvPlan(kind, dev_t(wk, dev), None if b is None else dev_t(b, dev), 2)
        got = plan(dev_t(x, dev))
        assert tuple(got.shape) == ref.shape == (n, 2 * h, 2 * w, 3)
        e_new, e_gen = rel_err(got.cpu().numpy(), ref), rel_err(gen(dev_t(x, dev)).cpu().numpy(), ref)
        assert e_new < 2e-6 and e_new < 4 * e_gen + 2e-7, (n, h, w, cin, e_new, e_gen)
        assert torch.equal(plan(dev_t(x[:1], dev)), got[:1])
        assert plan.flops(n, h, w) == gen.flops(n, h, w)
    # the padding origin, exactly: the 1-D integer pattern along each axis in channel 0, a delta at the other axis' index jc
    want = np.asarray(KERAS_UP if kind == "convT" else SIG_UP, np.float32)
    jc = 1 if kind == "convT" else 2
    for axis in (0, 1):
        x = np.zeros((1, 3, 3, 16), np.float32)
        w = np.zeros((5, 5, 3, 16) if kind == "convT" else (5, 5, 16, 3), np.float32)
        for a_, v in enumerate(SIG_X6[:3]):
            x[(0, a_, 0, 0) if axis == 0 else (0, 0, a_, 0)] = v
        for j, v in enumerate(SIG_W5):
            idx = (j, jc) if axis == 0 else (jc, j)
            for o in range(3):
                w[idx + ((o, 0) if kind == "convT" else (0, o))] = v * (o + 1)
        y = ops.UpSmallPlan(kind, dev_t(w, dev), None, 2)(dev_t(x, dev)).cpu().numpy()[0]          # [6, 6, 3]
        line = y[:, 0] if axis == 0 else y[0, :]
        np.testing.assert_array_equal(line, want[:, None] * np.array([1, 2, 3], np.float32)[None, :])
        rest = y[:, 1:] if axis == 0 else y[1:, :]
        assert not rest.any()
    # 9 x 9 / 4 (BLS2017Synthesis' last layer, reference common/transforms.py:131-134): four px phases per thread, a phase row per wave
    assert ops.UpSmallPlan.supported(kind, 9, 4, 256, 3) and not ops.UpSmallPlan.supported(kind, 9, 2, 256, 3)
    for n, h, w, cin in ((2, 16, 12, 256), (1, 5, 7, 32), (1, 19, 33, 64)):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wk = (rng.standard_normal((9, 9, 3, cin) if kind == "convT" else (9, 9, cin, 3)) / np.sqrt(81 * cin / 16)).astype(np.float32)
        b = rng.standard_normal(3).astype(np.float32)
        ref = fn(x, wk, b, 4)
        plan = ops.UpSmallPlan(kind, dev_t(wk, dev), dev_t(b, dev), 4)
        gen = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), 4)
        got = plan(dev_t(x, dev))
        assert tuple(got.shape) == ref.shape == (n, 4 * h, 4 * w, 3)
        e_new, e_gen = rel_err(got.cpu().numpy(), ref), rel_err(gen(dev_t(x, dev)).cpu().numpy(), ref)
        assert e_new < 2e-6 and e_new < 4 * e_gen + 2e-7, (n, h, w, cin, e_new, e_gen)
        assert torch.equal(plan(dev_t(x[:1], dev)), got[:1])
        assert plan.flops(n, h, w) == gen.flops(n, h, w)
    if kind == "sigup":
        tb = TR.BLS2017Synthesis(256)
        tb(dev_t(rng.standard_normal((1, 4, 4, 256)).astype(np.float32) * 0.3, dev))
        assert isinstance(tb._graph.layers[-1].plan, ops.UpSmallPlan) and tb._graph.layers[-1].plan.stride == 4
    t = TR.MBT2018Synthesis(192, output_channels=3) if kind == "sigup" else TR.CNNSynthesis(192, output_channels=3)
    t(dev_t(rng.standard_normal((1, 4, 4, 320)).astype(np.float32) * 0.3, dev))
    assert isinstance(t._graph.layers[-1].plan, ops.UpSmallPlan)


def test_rgb_first_layer_kernel_fuzz(dev):
    """Random sizes (1 ... 200 pixels a side, 1 ... 4 images): the first-layer kernel == the row-packed plan, bit for bit."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(2026)
    wk = (rng.standard_normal((5, 5, 3, 192)) * 0.2).astype(np.float32)
    b = rng.standard_normal(192).astype(np.float32)
    rp = ops.RowPackedConv(dev_t(wk, dev), dev_t(b, dev), 2, "leaky_relu")
    plan = ops.RgbConvPlan(dev_t(wk, dev), dev_t(b, dev), 2, "leaky_relu")
    for _ in range(24):
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 200)), int(rng.integers(1, 200))
        x = dev_t(rng.standard_normal((n, h, w, 3)).astype(np.float32), dev)
        assert torch.equal(plan(x), rp(x)), (n, h, w)


def test_errors_are_loud(dev):
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    with pytest.raises(ValueError):
        ops.pad_reflect(torch.zeros((1, 4, 4, 3)), 8, 8)                 # CPU tensor
    with pytest.raises(capi.SntcError):
        ops.pad_reflect(torch.zeros((1, 4, 4, 3), device=dev), 16, 16)   # reflect pad >= size
    with pytest.raises(capi.SntcError):
        ops.gdn_small(torch.zeros((1, 2, 2, 7), device=dev), torch.ones(7, device=dev), torch.zeros((7, 7), device=dev))
    w = torch.zeros((3, 3, 8, 16), device=dev)                            # kernel smaller than the stride
    with pytest.raises(capi.SntcError):
        ops.ConvPlan("convT", w, None, 4)


def _fuzz_cases(n=48, seed=2026):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        kind = rng.choice(["conv", "convT", "sigdown", "sigup"])
        if kind in ("conv", "sigdown"):
            k = int(rng.choice([1, 3, 5, 7, 9]))
            s = int(rng.choice([1, 2, 4]))
            h, w = int(rng.integers(1, 23)), int(rng.integers(1, 23))
        else:
            s = int(rng.choice([1, 2, 3, 4, 8]))
            k = int(rng.choice([c for c in (1, 3, 5, 6, 7, 9, 13) if c >= s and (kind == "convT" or c % 2 == 1)]))
            h, w = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        cin = int(rng.choice([1, 3, 8, 12, 24, 32, 33, 64, 96]))
        cout = int(rng.choice([1, 3, 5, 12, 24, 32, 40, 100]))
        cases.append((str(kind), k, s, cin, cout, int(rng.integers(1, 4)), h, w, rng.choice([None, "relu", "leaky_relu", "sigmoid"])))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "-".join(map(str, c)))
def test_conv_family_fuzz(case, dev):
    """Seeded random sweep over kernel / stride / channel counts (incl. channels that are not multiples of 4 or
    32 -> scalar gather + narrow epilogue paths), 1-pixel images and odd sizes."""
    from shallow_ntc_amd import ops
    kind, k, s, cin, cout, n, h, w, act = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wshape = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
    wk = (rng.standard_normal(wshape) / np.sqrt(max(k * k * cin / 4, 1))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    fn = {"conv": O.conv2d, "convT": O.conv2d_transpose, "sigdown": O.signal_conv_down, "sigup": O.signal_conv_up}[kind]
    ref = O.ACTIVATIONS[act](fn(x, wk, b, s))
    got = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s, act)(dev_t(x, dev)).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("n,h,w", [(2, 176, 200), (1, 161, 187), (2, 100, 120), (1, 512, 768), (1, 11, 40)])
def test_ms_ssim(n, h, w, dev):
    """tf.image.ssim / ssim_multiscale statistics (mshyper/models.py:321-331) against the float64 oracle,
    incl. odd sizes (symmetric padding before the 2x2 average pool) and the < 160 single-scale branch."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(h + w)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.clip(np.rint(np.stack([128 + 70 * np.sin(xx / 11 + c) * np.cos(yy / 6 - c) for c in range(3)], -1)[None]
                        + rng.normal(0, 6, size=(n, h, w, 3))), 0, 255).astype(np.float32)
    b = np.clip(np.rint(a + rng.normal(0, 9, size=a.shape)), 0, 255).astype(np.float32)
    ref, _ = O.image_quality(a, b)
    got = ops.image_quality(dev_t(a, dev), dev_t(b, dev))
    np.testing.assert_allclose(got, ref, rtol=3e-5)
    # pixels_float == the uint8 quantiser, kept as float
    x = (a / np.float32(255) - np.float32(0.5) + rng.normal(0, 0.01, size=a.shape)).astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 5), (0, 3), (0, 0)))
    np.testing.assert_array_equal(ops.pixels_float(dev_t(xp, dev), h, w).cpu().numpy(),
                                  O.floats_to_pixels(x, False).astype(np.float32))


@pytest.mark.parametrize("kind,k,s,cin,cout,h,w,act", [("conv", 5, 2, 320, 320, 16, 24, "relu"), ("conv", 5, 2, 480, 320, 32, 48, None),
                                                       ("convT", 5, 2, 320, 320, 8, 12, "relu"), ("conv", 3, 1, 256, 40, 6, 5, None)])
def test_split_k_layers(kind, k, s, cin, cout, h, w, act, dev):
    """Few output tiles per image + a long contraction -> deterministic split-K (workspace slabs added in a
    fixed order).  Same oracle tolerance, and a batch gives bit-identical rows to single-image calls."""
    from shallow_ntc_amd import _capi, ops
    rng = np.random.default_rng(k * cin + cout)
    wshape = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
    wk = (rng.standard_normal(wshape) / np.sqrt(k * k * cin / 4)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x = rng.standard_normal((3, h, w, cin)).astype(np.float32)
    plan = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s, act)
    assert _capi.load().sntc_conv_workspace_bytes(plan._h, 3, h, w) > 0          # these shapes do take the split path
    fn = {"conv": O.conv2d, "convT": O.conv2d_transpose}[kind]
    ref = O.ACTIVATIONS[act](fn(x, wk, b, s))
    xd = dev_t(x, dev)
    got = plan(xd)
    assert rel_err(got.cpu().numpy(), ref) < TOL
    for i in range(3):
        assert torch.equal(plan(xd[i:i + 1].contiguous()), got[i:i + 1])
    assert torch.equal(plan(xd), got)                                            # run-to-run deterministic
    with pytest.raises(_capi.SntcError):                                         # workspace is mandatory, never silently skipped
        _capi.call("sntc_conv_forward", plan._h, ops._ptr(xd), 3, h, w, ops._ptr(got), None, None, None, 0, ops._stream())


@pytest.mark.parametrize("ch,has_res,act", [(12, True, "igdn"), (24, False, "igdn"), (48, False, "relu")])
def test_two_layer_tail_with_fused_pixel_output(ch, has_res, act, dev):
    """The decoder form of the tail (crop + floats_to_pixels + quantize_image + integer SSE in the same launch) gives
    exactly the uint8 pixels and SSE of the float tail followed by the separate pixel kernel, incl. a cropped size."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(ch)
    n, hh, wh = 2, 21, 19
    t = (rng.standard_normal((n, hh, wh, ch * (2 if has_res else 1))) * 0.3).astype(np.float32)
    beta = (1 + 0.3 * rng.random(ch)).astype(np.float32)
    gamma = (0.05 * rng.random((ch, ch))).astype(np.float32)
    w2 = (rng.standard_normal((5, 5, 3, ch)) * 0.15).astype(np.float32)
    b2 = (rng.standard_normal(3) * 0.1).astype(np.float32)
    kind = ops.TAIL_ACTS[act]
    td, bd, gd, wd, b2d = (dev_t(a, dev) for a in (t, beta, gamma, w2, b2))
    recon = ops.two_layer_tail(td, ch, has_res, kind, bd, gd, wd, b2d)
    for (h, w) in ((2 * hh, 2 * wh), (2 * hh - 5, 2 * wh - 3)):
        ref = dev_t(rng.uniform(-0.5, 0.5, (n, h, w, 3)).astype(np.float32), dev)
        sse_want, px_want = ops.pixels_sse(ref, recon, want_pixels=True)
        px, sse = ops.two_layer_tail_pixels(td, ch, has_res, kind, bd, gd, wd, b2d, h, w, reference=ref)
        assert torch.equal(px, px_want) and torch.equal(sse, sse_want)
        px2, none = ops.two_layer_tail_pixels(td, ch, has_res, kind, bd, gd, wd, b2d, h, w)
        assert none is None and torch.equal(px2, px_want)
    assert 0 < int(px.min()) or int(px.max()) <= 255        # exercised, values are pixels


@pytest.mark.parametrize("kind,k,s,cin,cout", [("convT", 3, 1, 64, 96), ("convT", 5, 2, 32, 48), ("conv", 3, 1, 48, 64)])
def test_bf16x3_split_precision_plan(kind, k, s, cin, cout, dev):
    """The fenced experiment of DESIGN.md 8: operands split into three bfloat16 terms, six cross products on the bf16
    matrix cores, fp32 accumulation.  Against the float64 oracle it must be as accurate as the fp32 path (both ~1e-7
    relative), but it is NOT required to be bit-identical to it."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(k * 100 + cin)
    n, h, w = 2, 13, 17
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wk = (rng.standard_normal((k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = (O.conv2d_transpose if kind == "convT" else O.conv2d)(x, wk, b, s)
    p32 = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s)
    p3 = ops.ConvPlan(kind, dev_t(wk, dev), dev_t(b, dev), s, bf16x3=True)
    y32 = p32(dev_t(x, dev)).cpu().numpy()
    e32 = rel_err(y32, ref)
    for variant in (2, 4):
        p3.set_tile(variant)
        y3 = p3(dev_t(x, dev)).cpu().numpy()
        e3 = rel_err(y3, ref)
        assert e3 < 3e-6 and e3 < 4 * e32 + 1e-7, (variant, e3, e32)
    with pytest.raises(Exception):
        ops.ConvPlan("conv", dev_t(rng.standard_normal((5, 5, 3, 16)).astype(np.float32), dev), None, 2, bf16x3=True)   # Cin % 16 != 0


@pytest.mark.parametrize("k,s,cin,cout,n,h,w,act", [(5, 2, 3, 192, 2, 64, 96, None), (5, 2, 3, 64, 1, 37, 41, "relu"),
                                                  (5, 2, 3, 32, 3, 16, 18, "leaky_relu"), (3, 1, 4, 48, 1, 19, 23, None),
                                                  (5, 2, 1, 32, 1, 33, 20, None)])
def test_row_packed_first_layer(k, s, cin, cout, n, h, w, act, dev):
    """The RGB first layer as a row-packed plan (zero-pad once, one kernel row per 16-deep K stage, vector loads at 4-byte
    aligned addresses) == the float64 oracle's Keras SAME convolution, to the generic dword-gather path's own accuracy, for even
    and odd sizes (asymmetric SAME pads), and through the transforms that use it."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(k * 10 + cin + h)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wk = (rng.standard_normal((k, k, cin, cout)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, wk, b, s)
    if act == "relu":
        ref = O.relu(ref)
    elif act == "leaky_relu":
        ref = O.leaky_relu(ref)
    rp = ops.RowPackedConv(dev_t(wk, dev), dev_t(b, dev), s, act)
    gen = ops.ConvPlan("conv", dev_t(wk, dev), dev_t(b, dev), s, act)
    y = rp(dev_t(x, dev))
    assert tuple(y.shape) == ref.shape == tuple(gen(dev_t(x, dev)).shape)
    e_rp, e_gen = rel_err(y.cpu().numpy(), ref), rel_err(gen(dev_t(x, dev)).cpu().numpy(), ref)
    assert e_rp < 2e-6 and e_rp < 4 * e_gen + 2e-7, (e_rp, e_gen)
    alone = rp(dev_t(x[:1], dev))
    assert torch.equal(alone, y[:1])
    with pytest.raises(Exception):
        ops.ConvPlan("conv", dev_t(rng.standard_normal((5, 5, 4, 16)).astype(np.float32), dev), None, 2, rowpack=True)   # 5 * 4 > 16


@pytest.mark.parametrize("k,cout,n,h,w,act", [(5, 192, 2, 64, 96, None), (5, 192, 1, 37, 41, "relu"), (5, 128, 3, 16, 18, "leaky_relu"),
                                              (5, 256, 1, 33, 70, None), (3, 192, 1, 19, 23, None), (5, 192, 2, 130, 257, None),
                                              (4, 128, 1, 9, 5, "relu"), (5, 192, 1, 1, 1, None)])
def test_rgb_first_layer_kernel_is_bit_identical_to_the_row_packed_plan(k, cout, n, h, w, act, dev):
    """csrc/rgb_conv.hip (reference common/elic.py:147, common/transforms.py:183): the RGB first layer on its own kernel --
    weights resident in LDS, double-buffered input patch, SAME padding as zeros in the patch -- gives the SAME BITS as the
    row-packed gather-GEMM plan it replaces (the same k-ordered fma chains), for even and odd sizes (asymmetric pads), tiles that
    hang over the image, any number of persistent workgroups, an image alone or in a batch; and agrees with the float64 oracle's
    Keras SAME convolution to the row-packed path's own accuracy."""
    from shallow_ntc_amd import ops
    assert ops.RgbConvPlan.supported(k, 2, 3, cout, act)
    assert not ops.RgbConvPlan.supported(k, 1, 3, cout, act) and not ops.RgbConvPlan.supported(k, 2, 4, cout, act)
    assert not ops.RgbConvPlan.supported(5, 2, 3, 64, act) and not ops.RgbConvPlan.supported(5, 2, 3, cout, "sigmoid")
    rng = np.random.default_rng(k * 1000 + cout + h * 7 + w)
    x = rng.standard_normal((n, h, w, 3)).astype(np.float32)
    wk = (rng.standard_normal((k, k, 3, cout)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.ACTIVATIONS[act](O.conv2d(x, wk, b, 2))
    rp = ops.RowPackedConv(dev_t(wk, dev), dev_t(b, dev), 2, act)
    plan = ops.RgbConvPlan(dev_t(wk, dev), dev_t(b, dev), 2, act)
    xd = dev_t(x, dev)
    want = rp(xd)
    got = plan(xd)
    assert tuple(got.shape) == ref.shape and plan.out_hw(h, w) == ref.shape[1:3]
    assert torch.equal(got, want)
    assert rel_err(got.cpu().numpy(), ref) < 2e-6
    assert plan.flops(n, h, w) == rp.flops(n, h, w) == 2 * n * ref.shape[1] * ref.shape[2] * k * k * 3 * cout
    for wg in (1, 3, 7, 0):
        plan.set_workgroups(wg)
        assert torch.equal(plan(xd), want), wg
    assert torch.equal(plan(xd[:1].contiguous()), want[:1])
    # no bias; an in-place weight update
    nb = ops.RgbConvPlan(dev_t(wk, dev), None, 2, act)
    assert torch.equal(nb(xd), ops.RowPackedConv(dev_t(wk, dev), None, 2, act)(xd))
    wk2 = (wk * 0.5 + 0.01).astype(np.float32)
    plan.update(dev_t(wk2, dev), dev_t(b, dev))
    assert torch.equal(plan(xd), ops.RowPackedConv(dev_t(wk2, dev), dev_t(b, dev), 2, act)(xd))
    with pytest.raises(Exception):
        plan(xd, res=want)


def test_rgb_first_layer_kernel_inside_the_analysis_transform(dev):
    """ElicAnalysis (reduced widths do not qualify: 192 channels here) gives the same latents with the first layer on its own
    kernel and as the row-packed plan; empty batches pass through."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common import transforms as TR
    x = dev_t(np.random.default_rng(3).standard_normal((2, 64, 96, 3)).astype(np.float32) * 0.3, dev)
    outs = []
    for flag in (True, False):
        old = ops.RGB_FIRST_LAYER
        ops.RGB_FIRST_LAYER = flag
        try:
            t = TR.ElicAnalysis(channels=(192, 32, 32, 64))
            first = None
            y = t(x)
            first = t._graph.layers[0].plan
            assert isinstance(first, ops.RgbConvPlan if flag else ops.RowPackedConv)
            outs.append(y)
        finally:
            ops.RGB_FIRST_LAYER = old
    assert torch.equal(outs[0], outs[1])
    plan = ops.RgbConvPlan(dev_t(np.zeros((5, 5, 3, 192), np.float32), dev), None, 2)
    assert tuple(plan(torch.empty((0, 8, 8, 3), device=dev)).shape) == (0, 4, 4, 192)


@pytest.mark.gpu
def test_tuning_choices_travel_between_ranks_as_plain_data(dev):
    """ops.export_tuning() / import_tuning(): what rank 0 measured, applied by another rank's plans in creation order -- the
    same launches everywhere, identical bits (every candidate computes the same chains); plans that do not match are refused."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(5)
    w = dev_t((rng.standard_normal((3, 3, 96, 96)) * 0.05).astype(np.float32), dev)
    x = dev_t(rng.standard_normal((2, 40, 56, 96)).astype(np.float32), dev)
    base = len(ops._PLAN_REGISTRY)
    a = ops.ConvPlan("conv", w, None, 1, "relu")
    ref = a(x)
    a.tune(x)
    entries = [e for e in ops.export_tuning() if e[0] == base]
    assert len(entries) == 1 and entries[0][1:7] == ("conv", 96, 96, 2, 40, 56)
    # "another rank": a fresh plan of the same layer takes the choice under ITS index and computes the same bits
    b = ops.ConvPlan("conv", w, None, 1, "relu")
    ops.import_tuning([(base + 1,) + entries[0][1:]])
    assert b._tuned == a._tuned and torch.equal(b(x), ref)
    with pytest.raises(capi.SntcError, match="same plans"):
        ops.import_tuning([(base + 1, "conv", 192, 96) + entries[0][4:]])


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the hand-worked [DEP] known answers of tests/test_oracle_pins.py (SignalConv2D "same_zeros" origins, the sf / cdf pair
# selection of the noisy priors) put to the HIP kernels directly -- numbers derived in that file by scalar arithmetic, not by the
# oracle.  Integers below 2^24: the convolution results must be EXACT.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("axis", [0, 1])
def test_signal_conv_origins_hand_kats_on_the_gpu(axis, dev):
    from shallow_ntc_amd import ops
    from tests.test_oracle_pins import KERAS_DOWN, KERAS_UP, SIG_DOWN, SIG_UP, SIG_W5, SIG_X6
    c = 16                               # channel ch carries the pattern times (ch + 1) through a diagonal kernel

    def line(vals):
        a = np.asarray(vals, np.float32)[:, None] * np.arange(1, c + 1, dtype=np.float32)[None, :]
        return a.reshape((1, -1, 1, c) if axis == 0 else (1, 1, -1, c))

    def kernel(vals, other=None):
        """The 1-D pattern along ``axis`` through a diagonal (channel-wise) kernel; ``other``: a 5-tap delta at that index on the
        second axis (the transposed kinds need kernel >= stride on both axes)."""
        k = np.zeros((len(vals), c, c), np.float32)
        for j, v in enumerate(vals):
            k[j] = v * np.eye(c, dtype=np.float32)
        if other is None:
            return k.reshape((len(vals), 1, c, c) if axis == 0 else (1, len(vals), c, c))
        k2 = np.zeros((len(vals), len(vals), c, c), np.float32)
        if axis == 0:
            k2[:, other] = k
        else:
            k2[other, :] = k
        return k2

    def out_line(y):
        y = y.cpu().numpy()[0]
        ln, rest = (y[:, 0], y[:, 1:]) if axis == 0 else (y[0], y[1:])
        assert not rest.any()
        return ln

    scale = np.arange(1, c + 1, dtype=np.float32)[None, :]
    # second axis of the transposed kinds (input width 1 -> output width 2): SignalConv2D lands x[0] w[j2] on column j2 - 2, the
    # Keras layer on column j2 - 1: a delta at j2 = 2 / 1 puts the line on column 0 and nothing on column 1
    for kind, x, want, other in (("sigdown", SIG_X6, SIG_DOWN, None), ("conv", SIG_X6, KERAS_DOWN, None), ("sigup", SIG_X6[:3], SIG_UP, 2),
                                 ("convT", SIG_X6[:3], KERAS_UP, 1)):
        plan = ops.ConvPlan(kind, dev_t(kernel(SIG_W5, other), dev), None, 2)
        got = out_line(plan(dev_t(line(x), dev)))
        np.testing.assert_array_equal(got, np.asarray(want, np.float32)[:, None] * scale, err_msg=kind)


def test_prior_tail_selection_hand_kats_on_the_gpu(dev):
    """The closed forms of tests/test_oracle_pins.py: logistic prior P(v) = sigmoid(9 (v + .5)) - sigmoid(9 (v - .5)) and the
    normal far tails through the Mills series; fp32 kernels against float64 numbers: 2e-4 relative (the bpp bar is 1e-4 ABSOLUTE on
    sums over ~5e5 symbols whose typical term is a few bits)."""
    import math
    from shallow_ntc_amd import ops
    from tests.test_oracle_pins import _affine_prior, _log_sf_mills
    c = 4
    ms, bs, fs = _affine_prior(c, 0.0)
    prior = ops.DeepFactorizedPrior([m.astype(np.float32) for m in ms], [b.astype(np.float32) for b in bs],
                                    [f.astype(np.float32) for f in fs])
    want0 = -math.log2(math.tanh(2.25))
    want5 = (40.5 - math.log1p(-math.exp(-9.0)) + math.log1p(math.exp(-40.5))) / math.log(2)
    for v, want in ((0.0, want0), (5.0, want5), (-5.0, want5), (0.4, want0), (4.6, want5)):
        z_hat, bits = prior(dev_t(np.full((1, 1, 1, c), v), dev))
        assert float(z_hat.flatten()[0]) == round(v)
        assert abs(float(bits[0]) / c - want) <= 2e-4 * want, (v, float(bits[0]) / c, want)
    s0 = 0.11
    for v, lo, hi in ((20.0, 19.5 / s0, 20.5 / s0), (-3.0, 2.5 / s0, 3.5 / s0), (-20.0, 19.5 / s0, 20.5 / s0), (3.0, 2.5 / s0, 3.5 / s0)):
        ls_lo, ls_hi = _log_sf_mills(lo), _log_sf_mills(hi)
        want = -(ls_lo + math.log1p(-math.exp(ls_hi - ls_lo))) / math.log(2)
        y = np.full((1, 1, 1, c), v, np.float32)
        hyper = np.concatenate([np.zeros_like(y), np.full_like(y, -50.0)], -1)          # exp(raw) = 0 -> index 0 -> sigma 0.11
        _, bits, sym = ops.entropy_scale_normal(dev_t(y, dev), dev_t(hyper, dev), want_symbols=True)
        assert int(sym.flatten()[0]) == int(v)
        assert abs(float(bits[0]) / c - want) <= 2e-4 * want, (v, float(bits[0]) / c, want)

"""The fused first layer of the two-layer syntheses (csrc/syn_fused.hip; reference common/transforms.py:298-361) against
the layers it replaces -- bit for bit -- and against the float64 oracle (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu


def dev_t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def make_layer(rng, cin, ch, has_res, k=13):
    cp = ch * (2 if has_res else 1)
    w1 = (rng.standard_normal((k, k, cp, cin)) / np.sqrt(4 * cin)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(cp)).astype(np.float32)
    beta = (1.0 + rng.random(ch)).astype(np.float32)
    gamma = (0.1 * np.eye(ch) + 0.02 * rng.random((ch, ch))).astype(np.float32)
    return w1, b1, beta, gamma


def layered_hidden(ops, up, x, ch, has_res, kind, beta, gamma):
    """What the fused launch replaces: the phase-grouped transposed convolution, then stage 1 of the tail kernel."""
    return ops.two_layer_hidden(up(x), ch, has_res, kind, beta, gamma)


CASES = [  # cin, ch, has_res, act, [(n, h, w), ...]
    (320, 12, True, "igdn", [(1, 5, 7), (2, 16, 16), (1, 19, 23)]),
    (320, 12, True, "igdn", [(1, 32, 48), (1, 48, 32)]),
    (64, 24, False, "igdn", [(2, 9, 11), (1, 3, 127)]),
    (64, 12, False, "igdn", [(1, 17, 20), (3, 4, 4)]),
    (64, 48, False, "relu", [(1, 6, 37)]),
    (64, 24, True, "gdn", [(1, 10, 26)]),
    (32, 12, True, "lrelu", [(1, 1, 1), (1, 2, 300 // 3)]),
    (32, 12, True, None, [(2, 7, 5)]),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"cin{c[0]}-ch{c[1]}-res{int(c[2])}-{c[3]}")
def test_fused_synthesis_is_bit_identical_to_the_layers(case, dev):
    from shallow_ntc_amd import ops
    cin, ch, has_res, act, shapes = case
    rng = np.random.default_rng(cin + 7 * ch + has_res)
    w1, b1, beta, gamma = make_layer(rng, cin, ch, has_res)
    kind = ops.TAIL_ACTS[act]
    w1d, b1d, bd, gd = (dev_t(a, dev) for a in (w1, b1, beta, gamma))
    assert ops.SynPlan.supported(13, 8, cin, ch, has_res)
    syn = ops.SynPlan(w1d, b1d, 8, ch, has_res, kind, bd, gd)
    up = ops.ConvPlan("convT", w1d, b1d, 8)
    for (n, h, w) in shapes:
        x = dev_t(rng.standard_normal((n, h, w, cin)), dev)
        want = layered_hidden(ops, up, x, ch, has_res, kind, bd, gd)
        got = syn(x)
        assert got.shape == want.shape
        assert torch.equal(got, want), f"{(n, h, w)}: {int((got != want).sum())} of {got.numel()} values differ"
        for wg in (1, 3, 50):                      # any number of persistent workgroups: same items, same bits
            syn.set_workgroups(wg)
            assert torch.equal(syn(x), want)
        syn.set_workgroups(0)
    # against float64 (the first shape): conv-transpose, activation, residual
    n, h, w = shapes[0]
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    t = O.conv2d_transpose(x.astype(np.float64), w1, b1, 8)
    base = t[..., :ch]
    if act in ("igdn", "gdn"):
        base = O.gdn(base, beta, gamma, inverse=act == "igdn")
    elif act == "relu":
        base = O.relu(base)
    elif act == "lrelu":
        base = np.where(base >= 0, base, 0.2 * base)
    ref = base + (t[..., ch:] if has_res else 0.0)
    got = syn(dev_t(x, dev)).cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-5
    ops.check_conv_status()


def test_batches_of_different_sizes_share_one_launch(dev):
    """Up to four batches of different image sizes in one call (the Kodak set's two orientations): each batch's hidden tensor
    equals the one of a call of its own."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(5)
    w1, b1, beta, gamma = make_layer(rng, 320, 12, True)
    w1d, b1d, bd, gd = (dev_t(a, dev) for a in (w1, b1, beta, gamma))
    syn = ops.SynPlan(w1d, b1d, 8, 12, True, 1, bd, gd)
    xs = [dev_t(rng.standard_normal(s + (320,)), dev) for s in ((3, 32, 48), (2, 48, 32), (1, 5, 9), (2, 16, 16))]
    alone = [syn(x) for x in xs]
    together = syn(xs)
    for a, b in zip(alone, together):
        assert torch.equal(a, b)
    # repeated launches (the work queue is re-armed on the stream every call)
    for _ in range(5):
        for a, b in zip(alone, syn(xs)):
            assert torch.equal(a, b)


def test_weight_update_in_place(dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(11)
    w1, b1, beta, gamma = make_layer(rng, 64, 12, True)
    w1d, b1d, bd, gd = (dev_t(a, dev) for a in (w1, b1, beta, gamma))
    syn = ops.SynPlan(w1d, b1d, 8, 12, True, 1, bd, gd)
    x = dev_t(rng.standard_normal((2, 9, 13, 64)), dev)
    first = syn(x)
    w2, b2, beta2, gamma2 = make_layer(rng, 64, 12, True)
    w2d, b2d, b2eta, g2d = (dev_t(a, dev) for a in (w2, b2, beta2, gamma2))
    syn.update(w2d, b2d, b2eta, g2d)
    want = ops.SynPlan(w2d, b2d, 8, 12, True, 1, b2eta, g2d)(x)
    got = syn(x)
    assert torch.equal(got, want) and not torch.equal(got, first)


OTHER_FRAMES = [  # k, s, cin, ch, has_res, act, (n, h, w): every kernel / stride pair the nine-shift frame admits takes the run-time-shift loop
    (5, 2, 64, 12, True, "igdn", (2, 9, 14)),
    (5, 2, 32, 24, False, "relu", (1, 33, 5)),
    (3, 1, 48, 12, True, "igdn", (1, 21, 19)),
    (3, 1, 16, 48, False, None, (2, 8, 8)),
    (9, 4, 64, 12, True, "igdn", (1, 12, 17)),
    (9, 4, 32, 24, True, "lrelu", (3, 5, 6)),
    (4, 2, 32, 12, False, "igdn", (1, 10, 10)),
    (6, 4, 32, 12, True, "gdn", (1, 7, 13)),
    (16, 8, 32, 12, True, "igdn", (1, 6, 9)),
    (8, 8, 32, 24, False, "igdn", (1, 5, 8)),
    (2, 2, 16, 12, True, None, (2, 6, 7)),
    (1, 1, 32, 24, False, "relu", (1, 11, 13)),
    (18, 16, 32, 12, True, "igdn", (1, 4, 5)),
    (10, 8, 16, 48, False, "igdn", (1, 9, 4)),
]


@pytest.mark.parametrize("case", OTHER_FRAMES, ids=lambda c: f"k{c[0]}s{c[1]}-cin{c[2]}-ch{c[3]}-res{int(c[4])}-{c[5]}")
def test_other_kernel_and_stride_pairs(case, dev):
    """What ``SynPlan.supported`` accepts, a two-layer synthesis built with other ``kernel_sizes`` / ``strides`` runs: each such
    pair against the layers it replaces, bit for bit (pairs outside the frame are refused and keep the layers)."""
    from shallow_ntc_amd import ops
    k, s, cin, ch, has_res, act, (n, h, w) = case
    if not ops.SynPlan.supported(k, s, cin, ch, has_res):
        pytest.skip("outside the nine-shift frame: the layers stay")
    rng = np.random.default_rng(100 * k + s + cin)
    w1, b1, beta, gamma = make_layer(rng, cin, ch, has_res, k=k)
    kind = ops.TAIL_ACTS[act]
    w1d, b1d, bd, gd = (dev_t(a, dev) for a in (w1, b1, beta, gamma))
    syn = ops.SynPlan(w1d, b1d, s, ch, has_res, kind, bd, gd)
    up = ops.ConvPlan("convT", w1d, b1d, s)
    x = dev_t(rng.standard_normal((n, h, w, cin)), dev)
    want = layered_hidden(ops, up, x, ch, has_res, kind, bd, gd)
    got = syn(x)
    assert got.shape == want.shape == (n, h * s, w * s, ch)
    assert torch.equal(got, want), f"{int((got != want).sum())} of {got.numel()} values differ"
    ops.check_conv_status()


@pytest.mark.parametrize("cfg_name,hw", [("two_layer_syn", (96, 128)), ("two_layer_syn2", (70, 100))])
def test_decoded_pixels_do_not_change(cfg_name, hw, dev):
    """Model.decode with the fused synthesis == with the layered path: uint8 pixels and integer SSE, incl. a padded size."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **getattr(configs, cfg_name)(rd_lambda=0.01))
    rng = np.random.default_rng(3)
    x = dev_t(rng.uniform(-0.5, 0.5, (2,) + hw + (3,)), dev)
    z_hat, sym, _, _ = model.encode(x)
    assert ops.FUSED_SYNTHESIS
    px, sse = model.decode(z_hat, sym, hw, reference=x)
    ops.FUSED_SYNTHESIS = False
    try:
        px0, sse0 = model.decode(z_hat, sym, hw, reference=x)
    finally:
        ops.FUSED_SYNTHESIS = True
    assert torch.equal(px, px0) and torch.equal(sse, sse0)


def test_decode_set_equals_one_decode_per_batch(dev):
    """Model.decode_set (hyper-syntheses side by side, ONE synthesis launch for batches of different sizes, output layers per
    batch) returns exactly what Model.decode returns for each batch -- pixels and SSE; repeated calls included."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.01))
    rng = np.random.default_rng(8)
    codes = []
    for n, hw in ((3, (256, 384)), (2, (384, 256)), (1, (200, 120))):
        x = dev_t(rng.uniform(-0.5, 0.5, (n,) + hw + (3,)), dev)
        z_hat, sym, _, _ = model.encode(x)
        codes.append((z_hat, sym, hw, x))
    want = [model.decode(z, s, hw, reference=x) for z, s, hw, x in codes]
    old_min = ops.FUSED_SYNTHESIS_MIN_ITEMS
    ops.FUSED_SYNTHESIS_MIN_ITEMS = 1                     # small images: take the fused launch anyway
    try:
        for _ in range(3):
            got = model.decode_set(codes)
            for (px, sse), (px0, sse0) in zip(got, want):
                assert torch.equal(px, px0) and torch.equal(sse, sse0)
        plain = model.decode_set([c[:3] for c in codes])
        for px, (px0, _) in zip(plain, want):
            assert torch.equal(px, px0)
    finally:
        ops.FUSED_SYNTHESIS_MIN_ITEMS = old_min


def test_decode_replays_from_a_hip_graph(dev):
    """The decode sequence captured once (graphs.DecodeGraph) and replayed: EVERY replay returns the eager pixels.  The stream-K
    hand-off flags and the synthesis launch's work queue are re-armed by kernels in front of the launches that use them -- as
    memsets they were not re-armed on replay (the second replay differed, with or without the fused synthesis)."""
    from shallow_ntc_amd.graphs import DecodeGraph
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.01))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    n, hw = 6, (768, 512)
    z_hat = torch.round(3.0 * torch.randn((n, hw[0] // 64, hw[1] // 64, 320), device=dev, generator=g)).contiguous()
    sym = torch.round(2.0 * torch.randn((n, hw[0] // 16, hw[1] // 16, 320), device=dev, generator=g)).to(torch.int32).contiguous()
    want = model.decode(z_hat, sym, hw)
    graph = DecodeGraph(model, z_hat, sym, hw)
    for _ in range(4):
        assert torch.equal(graph(), want)
    sym2 = -sym                                            # new codes through the same graph
    want2 = model.decode(z_hat, sym2, hw)
    assert torch.equal(graph(z_hat, sym2), want2) and not torch.equal(want2, want)


def test_a_set_of_batches_replays_from_one_graph(dev):
    """graphs.DecodeSetGraph: both orientations as branches of ONE captured graph -- every replay equals the eager decodes."""
    from shallow_ntc_amd.graphs import DecodeSetGraph
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.01))
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    codes = []
    for n, hw in ((4, (512, 768)), (2, (768, 512))):
        z_hat = torch.round(3.0 * torch.randn((n, hw[0] // 64, hw[1] // 64, 320), device=dev, generator=g)).contiguous()
        sym = torch.round(2.0 * torch.randn((n, hw[0] // 16, hw[1] // 16, 320), device=dev, generator=g)).to(torch.int32).contiguous()
        codes.append((z_hat, sym, hw))
    want = [model.decode(z, s, hw) for z, s, hw in codes]
    graph = DecodeSetGraph(model, codes)
    for _ in range(4):
        for got, ref in zip(graph(), want):
            assert torch.equal(got, ref)
    flipped = [(z, -s, hw) for z, s, hw in codes]
    want2 = [model.decode(z, s, hw) for z, s, hw in flipped]
    for got, ref in zip(graph(flipped), want2):
        assert torch.equal(got, ref)


def test_fused_synthesis_is_taken_from_24_columns_per_phase_on(dev):
    """ops.FUSED_SYNTHESIS_MIN_COLUMNS: TwoLayerSynthesis(12, 3) (two_layer_syn2's default; 12 output columns per phase) decodes
    through the gather GEMM + the tail kernel's own activation stage (faster there: DESIGN.md 8), TwoLayerResSynthesis(12, 3) and
    TwoLayerSynthesis(24, 3) through the fused launch -- the same pixels either way (the fused kernel is bit-identical to the layers)."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common import transforms as TR
    rng = np.random.default_rng(12)
    y_hat = dev_t(rng.standard_normal((6, 48, 32, 320)).astype(np.float32), dev)           # 6 x 6 tiles x 8 units = 288 items
    old_items, old_cols = ops.FUSED_SYNTHESIS_MIN_ITEMS, ops.FUSED_SYNTHESIS_MIN_COLUMNS
    ops.FUSED_SYNTHESIS_MIN_ITEMS = 1
    try:
        for t, fused_by_default in ((TR.TwoLayerSynthesis(channels=(12, 3)), False), (TR.TwoLayerSynthesis(channels=(24, 3)), True),
                                    (TR.TwoLayerResSynthesis(channels=(12, 3)), True)):
            px, _ = t.forward_pixels(y_hat, 768, 512)
            assert t._use_syn(y_hat) == fused_by_default
            ops.FUSED_SYNTHESIS_MIN_COLUMNS = 12 if not fused_by_default else 1000       # the other path
            assert t._use_syn(y_hat) != fused_by_default
            px2, _ = t.forward_pixels(y_hat, 768, 512)
            ops.FUSED_SYNTHESIS_MIN_COLUMNS = old_cols
            assert torch.equal(px, px2)
    finally:
        ops.FUSED_SYNTHESIS_MIN_ITEMS, ops.FUSED_SYNTHESIS_MIN_COLUMNS = old_items, old_cols

"""Full-size (BASELINE.json configs 2-4 shapes) property tests on the GPU, where the float64 oracle
would take minutes: size-independent identities instead of element-wise comparison (-m gpu)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


@pytest.fixture(scope="module")
def kodak_model(dev):
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.02))
    w = dict(model.get_weights())
    rng = np.random.default_rng(0)
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[320:] = rng.uniform(-2.0, 2.5, size=320)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    model.set_weights(w)
    model._step = 10**9
    return model


def test_full_model_shapes_and_batch_invariance(kodak_model, dev):
    """y (n,32,48,320), z (n,8,12,320), x_hat (n,512,768,3) (get_flops.ipynb cell 26); a batch gives, image
    by image, bit-identical symbols / pixels to single-image evaluation (tile choice is speed only)."""
    from shallow_ntc_amd.common import data_lib
    m = kodak_model
    x = data_lib.normalize_image(data_lib.synthetic_images(3, 512, 768, seed=2))
    xd = t(x, dev)
    lat = m.infer_latent_rvs(xd)
    assert tuple(lat.uq[1].loc.shape) == (3, 32, 48, 320) and tuple(lat.uq[0].loc.shape) == (3, 8, 12, 320)
    z_hat, sym, bz, by = m.encode(xd)
    px, sse = m.decode(z_hat, sym, (512, 768), reference=xd)
    assert tuple(px.shape) == (3, 512, 768, 3) and px.dtype == torch.uint8
    for i in range(3):
        zi, si, bzi, byi = m.encode(xd[i:i + 1].contiguous())
        assert torch.equal(zi, z_hat[i:i + 1]) and torch.equal(si, sym[i:i + 1])
        assert abs(float(bzi[0]) - float(bz[i])) <= 1e-9 * abs(float(bz[i]))
        assert abs(float(byi[0]) - float(by[i])) <= 1e-9 * abs(float(by[i]))
        pi, ssei = m.decode(zi, si, (512, 768), reference=xd[i:i + 1].contiguous())
        assert torch.equal(pi, px[i:i + 1]) and int(ssei[0]) == int(sse[i])
    # decode is deterministic and the metrics follow the published identities
    px2, sse2 = m.decode(z_hat, sym, (512, 768), reference=xd)
    assert torch.equal(px, px2) and torch.equal(sse, sse2)
    rows = m.evaluate_batched(xd)
    for i, r in enumerate(rows):
        mse = int(sse[i]) / (512 * 768 * 3)
        assert abs(r["mse"] - mse) < 1e-3 * mse
        assert abs(r["psnr"] - 10 * np.log10(255.0 ** 2 / mse)) < 1e-3
        assert abs(r["rd_loss"] - (r["bpp"] + 0.02 * r["mse"])) < 1e-4 * r["rd_loss"]
        assert abs(r["bpp"] - (float(bz[i]) + float(by[i])) / (512 * 768)) < 1e-6 * r["bpp"]
    # the integer checksum of the decoded pixels equals the one of the evaluation path
    assert int(px.to(torch.int64).sum()) == int(m.decode(z_hat, sym, (512, 768)).to(torch.int64).sum())


def test_portrait_and_padded_sizes(kodak_model, dev):
    """768 x 512 (Kodak portrait) and a 500 x 750 image that reflect-pads to 512 x 768."""
    from shallow_ntc_amd.common import data_lib
    m = kodak_model
    xp = t(data_lib.normalize_image(data_lib.synthetic_images(1, 768, 512, seed=3)), dev)
    z_hat, sym, _, _ = m.encode(xp)
    assert tuple(sym.shape) == (1, 48, 32, 320) and tuple(m.decode(z_hat, sym, (768, 512)).shape) == (1, 768, 512, 3)
    xo = t(data_lib.normalize_image(data_lib.synthetic_images(1, 500, 750, seed=4)), dev)
    z_hat, sym, _, _ = m.encode(xo)
    assert tuple(sym.shape) == (1, 32, 48, 320)
    px = m.decode(z_hat, sym, (500, 750))
    assert tuple(px.shape) == (1, 500, 750, 3)
    met = next(iter(m.evaluate(xo))).scalars_float
    assert np.isfinite(met["bpp"]) and np.isfinite(met["psnr"])


def test_evaluate_groups_same_shaped_images_without_changing_a_number(kodak_model, dev):
    """Model.evaluate() launches same-shaped images of its look-ahead window together (Kodak: landscape and portrait
    interleaved); every image's metrics -- bpp, PSNR, MS-SSIM, rd_loss -- and its reconstruction are those of the strictly
    serial one-image-per-pass loop of the reference (mshyper/models.py:415-433), in input order."""
    from shallow_ntc_amd.common import data_lib
    m = kodak_model
    shapes = [(512, 768), (512, 768), (768, 512), (512, 768), (768, 512), (512, 768), (512, 768)]
    imgs = [t(data_lib.normalize_image(data_lib.synthetic_images(1, h, w, seed=20 + i)), dev) for i, (h, w) in enumerate(shapes)]
    keep = m._quality_metrics
    m._quality_metrics = True
    try:
        serial = list(m.evaluate(imgs, lookahead=1))
        grouped = list(m.evaluate(iter(imgs), lookahead=2, group=4))
    finally:
        m._quality_metrics = keep
    assert len(grouped) == len(imgs) and "msssim" in serial[0].scalars_float
    for a, b, (h, w) in zip(serial, grouped, shapes):
        assert a.scalars_float == b.scalars_float
        assert tuple(b.images["reconstruction"].shape) == (1, h, w, 3)
        assert torch.equal(a.images["reconstruction"], b.images["reconstruction"])


@pytest.mark.parametrize("kind,k,s,cin,cout,h,w", [("convT", 13, 8, 320, 24, 32, 48), ("convT", 5, 2, 320, 480, 16, 24),
                                                   ("convT", 3, 1, 480, 640, 32, 48), ("convT", 18, 16, 320, 3, 32, 48),
                                                   ("conv", 5, 2, 192, 192, 128, 192), ("conv", 3, 1, 96, 96, 64, 96)])
def test_linearity_and_adjoint_at_full_width(kind, k, s, cin, cout, h, w, dev):
    """T(a u + b v) = a T(u) + b T(v), and <T(u), g> = <u, T*(g)> where T* is the SAME conv / transposed conv
    on the same kernel array (the identity SGA's backward relies on)."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(k * 7 + s)
    wshape = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
    wk = t((rng.standard_normal(wshape) / np.sqrt(k * k * cin / 4)), dev)
    plan = ops.ConvPlan(kind, wk, None, s)
    u, v = t(rng.standard_normal((2, h, w, cin)), dev), t(rng.standard_normal((2, h, w, cin)), dev)
    a, b = 0.75, -1.5
    lhs = plan((a * u + b * v).contiguous())
    rhs = a * plan(u) + b * plan(v)
    assert float((lhs - rhs).abs().max()) <= 2e-5 * float(rhs.abs().max())
    if kind == "convT":      # adjoint: Conv2D SAME stride s with the same array read as HWIO (I = cout, O = cin)
        adj = ops.ConvPlan("conv", wk, None, s)
        tu = plan(u)
        g = t(rng.standard_normal(tuple(tu.shape)), dev)
        left = float((tu.double() * g.double()).sum())
        right = float((u.double() * adj(g).double()).sum())
        assert abs(left - right) <= 1e-5 * (abs(left) + float(tu.abs().mean()) * float(g.abs().mean()) * tu.numel() ** 0.5)


def test_zero_input_is_bias_full_size(dev):
    """vis_syn_filters.ipynb: JPEG-like synthesis (k = 18, s = 16, 320 channels) of zeros is its bias."""
    from shallow_ntc_amd.common.transforms import class_builder
    tr = class_builder.build("JPEGLikeSynthesis", kernel_size=18, strides=16)
    tr.build(320, dev)
    w = dict(tr.get_weights())
    w["conv/bias"] = np.array([-0.00739, -0.04296, -0.08146], np.float32)
    tr.set_weights(w)
    out = tr(torch.zeros((2, 32, 48, 320), device=dev)).cpu().numpy()
    assert out.shape == (2, 512, 768, 3)
    np.testing.assert_array_equal(out, np.broadcast_to(w["conv/bias"], out.shape))


def test_batches_beyond_the_32bit_offset_limit_are_split(dev):
    """Inputs of >= 2 GiB are processed in halves (ops.ConvPlan); the result equals per-image calls."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(1)
    wk = t(rng.standard_normal((1, 1, 192, 32)) * 0.1, dev)
    plan = ops.ConvPlan("conv", wk, None, 1, "relu")
    x = torch.randn((8, 512, 768, 192), device=dev)          # 2.4 GB
    assert x.numel() * 4 >= ops.MAX_INPUT_BYTES
    y = plan(x)
    for i in (0, 3, 7):
        assert torch.equal(y[i:i + 1], plan(x[i:i + 1].contiguous()))
    from shallow_ntc_amd import _capi
    with pytest.raises(_capi.SntcError):                        # the C ABI itself refuses instead of wrapping around
        _capi.call("sntc_conv_forward", plan._h, ops._ptr(x), 8, 512, 768, ops._ptr(y), None, None, None, 0, ops._stream())


def test_edge_shapes(dev):
    from shallow_ntc_amd import _capi, ops
    rng = np.random.default_rng(2)
    wk = t(rng.standard_normal((5, 5, 32, 32)) * 0.1, dev)
    plan = ops.ConvPlan("conv", wk, None, 2)
    assert tuple(plan(torch.randn((1, 1, 1, 32), device=dev)).shape) == (1, 1, 1, 32)       # 1 x 1 image
    assert tuple(plan(torch.randn((3, 2, 7, 32), device=dev)).shape) == (3, 1, 4, 32)
    with pytest.raises(_capi.SntcError):
        plan(torch.zeros((0, 4, 4, 32), device=dev))                                        # empty batch
    with pytest.raises(ValueError):
        plan(torch.zeros((1, 4, 4, 16), device=dev))                                        # wrong channels


BASELINE_CONFIGS = [  # BASELINE.json configs 1-5 at their real widths; (name, factorized, batch, h, w)
    ("bls2017", True, 1, 256, 256),
    ("mbt2018", False, 8, 256, 256),
    ("jpegl", False, 2, 512, 768),
    ("two_layer_syn2", False, 1, 1200, 1200),     # Tecnick: pads to 1216 x 1216
]


@pytest.mark.parametrize("name,factorized,n,h,w", BASELINE_CONFIGS, ids=[c[0] for c in BASELINE_CONFIGS])
def test_every_baseline_config_runs_at_full_size(name, factorized, n, h, w, dev):
    """Each reference config builds from its dict, evaluates a full-size batch, and satisfies the metric
    identities; decode(encode(x)) reproduces the evaluation's distortion exactly (integer SSE)."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.factorized.models import Model as FModel
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    cfg = configs.CONFIGS[name](rd_lambda=0.02)
    model = (FModel if factorized else Model)(device=dev, **cfg)
    model._step = 10**9
    x = t(data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=11)), dev)
    rows = model.evaluate_batched(x)
    assert len(rows) == n
    codes = model.encode(x)
    px, sse = model.decode(codes[0], codes[1], (h, w), reference=x)
    assert tuple(px.shape) == (n, h, w, 3)
    for i, r in enumerate(rows):
        assert np.isfinite(r["bpp"]) and r["bpp"] > 0 and np.isfinite(r["psnr"])
        assert abs(r["rd_loss"] - (r["bpp"] + 0.02 * r["mse"])) <= 1e-4 * r["rd_loss"]
        assert abs(r["mse"] - int(sse[i]) / (h * w * 3)) <= 1e-4 * r["mse"]
    expected_factor = 16 if factorized else 64
    assert model.downsample_factor == expected_factor


def test_sga_step_at_tecnick_shape(dev):
    """BASELINE config 5: two_layer_syn2 (hidden 24) + itinf overrides, one 1200 x 1200 image, three SGA steps."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    cfg = {**configs.two_layer_syn2(rd_lambda=0.02, hidden_channels=24), **configs.itinf()}
    model = Model(device=dev, **cfg)
    x = t(data_lib.normalize_image(data_lib.synthetic_images(1, 1200, 1200, seed=12)), dev)
    model.initialize_itinf(x)
    assert tuple(model.latent_rvs.uq[1].loc.shape) == (1, 76, 76, 320) and tuple(model.latent_rvs.uq[0].loc.shape) == (1, 19, 19, 320)
    first = model.itinf_train_step(x, seed=1).scalars_float
    for _ in range(2):
        last = model.itinf_train_step(x, seed=1).scalars_float
    assert np.isfinite(first["rd_loss"]) and np.isfinite(last["rd_loss"]) and model.global_step == 3
    val = model.itinf_validation_step(x).scalars_float
    assert np.isfinite(val["psnr"])


@pytest.mark.parametrize("name,shape", [("two_layer_syn", (1, 512, 768)), ("mbt2018", (2, 256, 256)), ("jpegl", (1, 768, 512)),
                                        ("two_layer_syn2", (1, 256, 320))])
def test_full_width_models_against_the_independent_cpu_restatement(name, shape, dev):
    """Full-width reference configs (BASELINE.json configs 2-5) at full-size inputs: the HIP analysis output and the decoded
    pixels against oracle/torch_ref.py (float32 PyTorch-CPU, library convolutions -- an implementation that shares nothing
    with the gather-GEMM).  Tolerances: latents 1e-4 of their range (different fp32 summation orders over K up to 4800),
    decoded u8 pixels equal except for values on a rounding boundary (<= 1 code value on < 1e-3 of the pixels)."""
    from oracle import model_np, torch_ref
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    cfg = configs.CONFIGS[name]()
    model = Model(device=dev, quality_metrics=False, **cfg)
    w = model.get_weights()
    ref_model = model_np.Model(cfg["transform_config"], rd_lambda=cfg["rd_lambda"])
    n, h, wd = shape
    x = data_lib.normalize_image(data_lib.synthetic_images(n, h, wd, seed=12))
    lat = model.infer_latent_rvs(x)
    y_ref = torch_ref.to_nhwc(torch_ref.analysis_only(ref_model, w, x))
    y = lat.uq[1].loc.cpu().numpy()
    assert y.shape == y_ref.shape == (n, h // 16, wd // 16, y.shape[-1])
    assert np.abs(y - y_ref).max() < 1e-4 * np.abs(y_ref).max()
    z_hat, sym, _, _ = model.encode(x)
    px = model.decode(z_hat, sym, (h, wd)).cpu().numpy()
    px_ref = torch_ref.decode(ref_model, w, z_hat.cpu().numpy(), sym.cpu().numpy().astype(np.float32), (h, wd))
    diff = np.abs(px.astype(np.int16) - px_ref.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (int(diff.max()), float((diff > 0).mean()))
    # rate: the float64 entropy oracle on the SAME latents / hyper-synthesis output (north star: <= 1e-4 bpp, integer
    # symbols bit-exact), and PSNR of the two pixel sets against the input within 1e-3 dB
    from oracle import model_np as M
    from oracle import ops_np as O
    hyper = model._hyper_synthesis(z_hat).cpu().numpy().astype(np.float64)
    c = hyper.shape[-1] // 2
    _, bits_y_ref, sym_ref = O.scale_indexed_normal(y.astype(np.float64), hyper[..., :c], np.exp(hyper[..., c:]))
    ms, bs, fs = M._prior_lists(w)
    zq_ref, bits_z_ref = O.batched_deep_factorized(lat.uq[0].loc.cpu().numpy().astype(np.float64), ms, bs, fs)
    _, _, bits_z, bits_y = model.encode(x)
    np.testing.assert_array_equal(sym.cpu().numpy(), np.asarray(sym_ref, np.int64))
    np.testing.assert_array_equal(z_hat.cpu().numpy(), zq_ref)
    npix = float(h * wd)
    assert np.abs(bits_y.cpu().numpy() - bits_y_ref).max() / npix < 1e-4 and np.abs(bits_z.cpu().numpy() - bits_z_ref).max() / npix < 1e-4
    x255 = np.rint((x.astype(np.float64) + 0.5) * 255.0)
    psnr = lambda p: 10.0 * np.log10(255.0 ** 2 / ((x255 - p.astype(np.float64)) ** 2).mean(axis=(1, 2, 3)))
    assert np.abs(psnr(px) - psnr(px_ref)).max() < 1e-3


@pytest.mark.parametrize("kind,k,s,cin,cout,n,h,w,epi", [
    ("convT", 3, 1, 480, 640, 18, 32, 48, False),      # 2160 tiles of 270 stages: the decode's largest layer
    ("convT", 5, 2, 320, 480, 18, 16, 24, False),      # four phase groups, 7.5 column tiles each
    ("convT", 13, 8, 320, 24, 18, 32, 48, False),      # groups with 40 / 20 / 20 / 10 stages per tile
    ("conv", 5, 2, 192, 192, 6, 128, 192, False),
    ("conv", 1, 1, 96, 192, 6, 128, 192, True),        # short K (6 stages) + residual epilogue
    ("conv", 5, 2, 3, 192, 4, 256, 384, False),        # the RGB first layer: dword gather path
])
def test_stream_k_is_bit_identical_to_the_static_schedule(kind, k, s, cin, cout, n, h, w, epi, dev):
    """Persistent stream-K workers cut tiles between workgroups and CONTINUE the fma chains (csrc/gather_gemm.hip), so
    the result must equal the one-workgroup-per-tile schedule bit for bit, for every tile shape, and an image must
    come out the same alone as inside the batch."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(k * 1000 + cin)
    x = torch.randn((n, h, w, cin), device=dev, generator=g)
    wshape = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
    wk = torch.randn(wshape, device=dev, generator=g) * 0.05
    b = torch.randn((cout,), device=dev, generator=g)
    plan = ops.ConvPlan(kind, wk, b, s, "relu" if epi else None, capi.PRO_NONE, capi.EPI_ADD if epi else capi.EPI_STORE)
    ho, wo = plan.out_hw(h, w)
    res = torch.randn((n, ho, wo, cout), device=dev, generator=g) if epi else None
    plan.set_stream_k(True, force=True)        # short-tile shapes default to one workgroup per tile since round 3: force the cut
    v_auto, blocks = plan.launch_info(n, h, w)
    y_sk = plan(x, res=res).clone()
    plan.set_stream_k(False)
    _, blocks_static = plan.launch_info(n, h, w)
    y_static = plan(x, res=res).clone()
    assert blocks_static > blocks, "the shape was meant to run on the persistent workers"      # stream-K really ran
    assert torch.equal(y_sk, y_static)
    plan.set_stream_k(True, force=True)
    for variant in (1, 2, 3, 4, 5, 8, 9, 10):
        plan.set_tile(variant)
        assert torch.equal(plan(x, res=res), y_static), variant
    # the two stage paths (registers + ds_write / direct-to-LDS buffer loads), under both schedules
    for variant in (1, 2, 3, 4, 5, 8, 9):
        plan.set_tile(variant)
        for sk in (True, False):
            plan.set_stream_k(sk, dma=True, force=sk)
            assert torch.equal(plan(x, res=res), y_static), (variant, sk, "dma")
            plan.set_stream_k(sk, dma=False, force=sk)
            assert torch.equal(plan(x, res=res), y_static), (variant, sk, "registers")
    # the stream-K unit order: column tile outermost (the COLM twin of the 128 x 128 instance; single-group plans) against
    # row strip outermost -- other K ranges per worker, the same chains
    plan.set_tile(9)
    orders = set()
    for colm in (True, False):
        plan.set_stream_k(True, force=True, colm=colm)
        order = C.c_int(-1)
        capi.call("sntc_conv_launch_order", plan._h, n, h, w, C.byref(order))
        orders.add(order.value)
        assert torch.equal(plan(x, res=res), y_static), ("column-major", colm)
    if (kind, k, cin) == ("convT", 3, 480):  # one phase group on the vector path: the twin exists and was really launched
        assert orders == {0, 1}, orders
    plan.set_stream_k(True)
    plan.set_tile(0)
    one = plan(x[2:3].contiguous(), res=None if res is None else res[2:3].contiguous())
    assert torch.equal(one, y_static[2:3])
    torch.cuda.synchronize()


def test_concurrent_stream_k_launches_and_the_status_word(dev):
    """Stream-K needs every worker of a launch resident; four streams each launching stream-K layers oversubscribe the device
    (the advisor's scenario).  The launches must complete, give the bits of the static schedule, and leave the sticky status
    word clear; a flagged hand-off would raise in check_conv_status instead of trapping the context.  The process-wide switch
    puts every later call on the static schedule (fewer, smaller launches: more blocks than resident workers)."""
    from shallow_ntc_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    layers = []
    for kind, k, s, cin, cout, n, h, w in [("convT", 3, 1, 480, 640, 6, 32, 48), ("convT", 5, 2, 320, 480, 6, 16, 24),
                                          ("conv", 5, 2, 192, 192, 4, 128, 192), ("convT", 3, 1, 480, 640, 4, 32, 48)]:
        x = torch.randn((n, h, w, cin), device=dev, generator=g)
        wk = torch.randn((k, k, cout, cin) if kind == "convT" else (k, k, cin, cout), device=dev, generator=g) * 0.05
        plan = ops.ConvPlan(kind, wk, None, s)
        plan.set_stream_k(False)
        ref = plan(x).clone()
        plan.set_stream_k(True, force=True)
        _, blocks_sk = plan.launch_info(n, h, w)
        layers.append((plan, x, ref, blocks_sk))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    outs = []
    for rep in range(6):
        for (plan, x, ref, _), st in zip(layers, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append((plan(x), ref))
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    ops.check_conv_status()                         # no hand-off timed out
    for y, ref in outs:
        assert torch.equal(y, ref)
    try:
        ops.set_stream_k(False)
        changed = 0
        for plan, x, ref, blocks_sk in layers:
            n, h, w = x.shape[:3]
            _, blocks = plan.launch_info(n, h, w)
            changed += blocks != blocks_sk           # one workgroup per tile now (another tile shape may be picked with it)
            assert torch.equal(plan(x), ref)
        assert changed >= 2
    finally:
        ops.set_stream_k(True)


@pytest.mark.gpu
def test_tuned_schedule_is_a_speed_choice_only(dev):
    """sntc_conv_plan_tune measures the (tile, schedule) candidates of a plan for one call shape and records the fastest; the
    bits do not change (every candidate is the same k-ordered chain), other shapes and forced tiles are unaffected, and the
    entry can be cleared."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    for kind, k, s, cin, cout, shape in (("convT", 5, 2, 320, 480, (6, 24, 16)), ("conv", 3, 1, 96, 96, (2, 64, 96)),
                                         ("conv", 1, 1, 192, 96, (3, 64, 64))):
        wshape = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
        wk = torch.randn(wshape, device=dev, generator=g) * 0.05
        b = torch.randn((cout,), device=dev, generator=g)
        x = torch.randn(shape + (cin,), device=dev, generator=g)
        p = ops.ConvPlan(kind, wk, b, s, "relu")
        before = p(x).clone()
        info0 = p.launch_info(*shape)
        other = x[:1].contiguous()
        other_info = p.launch_info(1, shape[1], shape[2])
        v, sk = p.tune(x)
        assert 1 <= v <= 10 and sk in (0, 1)
        assert p.launch_info(*shape)[0] == v
        assert p.launch_info(1, shape[1], shape[2]) == other_info          # another batch size: untouched
        assert torch.equal(p(x), before)
        assert torch.equal(p(other), before[:1])
        p.set_tile(2)                                                        # a forced tile wins over the tuned entry
        assert p.launch_info(*shape)[0] == 2 and torch.equal(p(x), before)
        p.set_tile(0)
        with ops.autotune(reps=2):                                           # the context manager tunes unseen shapes on the fly
            assert torch.equal(p(other), before[:1])
        assert (1, shape[1], shape[2]) in p._tuned
        p.clear_tuning()
        assert p.launch_info(*shape) == info0
        # the candidate list, and a choice recorded from outside (ops.tune_step does this from a whole step's clock)
        cands = p.candidates(*shape)
        assert (v, sk) in cands and len(cands) >= 2
        ov, osk = next(c for c in cands if c != (v, sk))
        p.set_choice(*shape, ov, osk)
        assert p.launch_info(*shape)[0] == ov and torch.equal(p(x), before)
        with pytest.raises(capi.SntcError):
            p.set_choice(*shape, 10, 0)                                      # 256 x 128 is never a candidate
        p.clear_tuning()
    # a whole step on two streams chosen by its own clock: same bits whatever it settles on
    pa = ops.ConvPlan("convT", torch.randn((3, 3, 64, 64), device=dev, generator=g) * 0.05, None, 1, "relu")
    pb = ops.ConvPlan("conv", torch.randn((3, 3, 64, 64), device=dev, generator=g) * 0.05, None, 1)
    xa = torch.randn((6, 32, 48, 64), device=dev, generator=g)
    xb = torch.randn((2, 48, 32, 64), device=dev, generator=g)
    want = (pa(xa).clone(), pb(xb).clone())
    side = torch.cuda.Stream(device=dev)
    outs = {}

    def step():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            outs["b"] = pb(xb)
        outs["a"] = pa(xa)
        cur.wait_stream(side)

    log = []
    t0, t1 = ops.tune_step(step, reps=4, log=log)
    assert t0 > 0 and t1 > 0 and len(log) == 2
    step()
    torch.cuda.synchronize()
    assert torch.equal(outs["a"], want[0]) and torch.equal(outs["b"], want[1])
    # samples of several steps back to back, the launches walked again while a walk changed something: same bits; a launch
    # no candidate improves keeps the choice it had
    had = {id(p): dict(p._tuned) for p in (pa, pb)}
    log2 = []
    t2, t3 = ops.tune_step(step, reps=3, burst=3, passes=2, min_gain=0.5, log=log2)      # nothing gains 50 %: every choice stays
    assert t2 > 0 and t3 > 0 and all(r.get("chosen") is None for r in log2) and {id(p): dict(p._tuned) for p in (pa, pb)} == had
    step()
    torch.cuda.synchronize()
    assert torch.equal(outs["a"], want[0]) and torch.equal(outs["b"], want[1])
    # a step whose stream-K hand-offs time out (the status word, injected): no candidate is taken from such samples, the
    # launches keep what they came with, and the call ends like check_conv_status -- stream-K off, an error
    from shallow_ntc_amd import _capi as capi_mod
    calls = {"n": 0, "from": 0}

    def flagged_step():
        step()
        calls["n"] += 1
        if calls["n"] > calls["from"]:
            capi_mod.call("sntc_conv_status_inject", 1, ops._stream())

    for clean_calls, message in ((4, "timed out with the schedules it came with"),     # flagged from the first sample on
                                 (8, "hand-offs of the step time out")):              # ... from the candidates on: none is taken
        calls.update(n=0)
        calls["from"] = clean_calls
        with pytest.raises(capi_mod.SntcError, match=message):
            ops.tune_step(flagged_step, reps=2, log=[])
        assert not ops.stream_k_enabled() and {id(p): dict(p._tuned) for p in (pa, pb)} == had
        # the time-out latched stream-K off for the process: restoring an earlier setting does not re-arm it ...
        ops.set_stream_k(True)
        assert not ops.stream_k_enabled()
        with ops.static_schedules():
            pass
        assert not ops.stream_k_enabled()
        ops.set_stream_k(True, force=True)       # ... only a caller that knows the device is its own again does
        ops.take_conv_status()
    assert ops.stream_k_enabled() and ops.take_conv_status() == 0
    step()
    torch.cuda.synchronize()
    assert torch.equal(outs["a"], want[0]) and torch.equal(outs["b"], want[1])
    # a pre-split bf16 x 3 plan tunes over its own two tiles
    wk = torch.randn((3, 3, 64, 64), device=dev, generator=g) * 0.05
    x = torch.randn((4, 32, 48, 64), device=dev, generator=g)
    ps = ops.ConvPlan("convT", wk, None, 1, bf16x3="presplit")
    xs = ops.split3(x)
    before = ps(xs).clone()
    v, sk = ps.tune(xs)
    assert v in (11, 12) and torch.equal(ps(xs), before)
    torch.cuda.synchronize()
    ops.check_conv_status()


def test_counted_flops_are_the_reference_tables(kodak_model, dev):
    """What bench.py divides by the measured time (ops.PROFILE: the algorithmic 2 * MAC count of every convolution launch) is
    the reference's own FLOPs per pixel (results/all_fpp.csv:3-4, tests/golden/published_rows.json) times the pixels: decode
    g_h + g = 41,031.7 (the TF profiler's figure includes bias / activation adds: within 1 %), encode f + f_h + g_h =
    510,564 + 13,452 + 30,355; and the SGA adjoint of the synthesis counts its 24 real gradient channels, not the 32 it pads to."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common import data_lib
    m = kodak_model
    x = t(data_lib.normalize_image(data_lib.synthetic_images(2, 512, 768, seed=5)), dev)
    px = 2 * 512 * 768
    ops.PROFILE = []
    try:
        z_hat, sym, _, _ = m.encode(x)
        enc = sum(e["flops"] for e in ops.PROFILE)
        ops.PROFILE = []
        m.decode(z_hat, sym, (512, 768))
        dec = sum(e["flops"] for e in ops.PROFILE)
    finally:
        ops.PROFILE = None
    assert abs(dec / px / 41031.69 - 1) < 0.01
    assert abs(enc / px / (510563.75 + 13451.64 + 30354.69) - 1) < 0.01
    from shallow_ntc_amd.sga import TwoLayerBackward
    bw = TwoLayerBackward(m._synthesis)
    assert bw.up_adj.cin == 32 and bw.up_adj.flops(1, 256, 384) == 2 * 32 * 48 * 169 * 24 * 320      # 24 real channels, padded to 32


def test_static_schedules_switch_off_stream_k_and_restore_it(dev):
    """``with ops.static_schedules():`` -- the launches inside leave the persistent stream-K workers alone (what
    ``decompress_many`` asks of the convolutions that run beside entropy-decoding waves); same bits, the switch comes back."""
    from shallow_ntc_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    p = ops.ConvPlan("convT", torch.randn((3, 3, 256, 480), device=dev, generator=g) * 0.02, None, 1)
    x = torch.randn((18, 32, 48, 480), device=dev, generator=g)
    shape = tuple(x.shape[:3])
    info = p.launch_info(*shape)
    want = p(x).clone()
    assert ops.stream_k_enabled()
    with ops.static_schedules():
        assert not ops.stream_k_enabled()
        inside = p.launch_info(*shape)
        got = p(x).clone()
        with ops.static_schedules():                     # nested: the inner block leaves the switch as it found it
            assert not ops.stream_k_enabled()
        assert not ops.stream_k_enabled()
    assert ops.stream_k_enabled() and p.launch_info(*shape) == info
    # a hand-off that times out INSIDE the block: check_conv_status switches stream-K off for the process, and leaving the block
    # must not switch it back on
    from shallow_ntc_amd import _capi as capi_mod
    try:
        with ops.static_schedules():
            capi_mod.call("sntc_conv_status_inject", 1, ops._stream())
            with pytest.raises(capi_mod.SntcError, match="stream-K"):
                ops.check_conv_status()
        assert not ops.stream_k_enabled()
    finally:
        ops.set_stream_k(True, force=True)
    assert ops.stream_k_enabled()
    assert inside != info, "the launch was expected to be a stream-K one outside the block"
    assert torch.equal(got, want)
    with ops.static_schedules(False):                    # inactive: nothing changes
        assert ops.stream_k_enabled() and p.launch_info(*shape) == info
    ops.check_conv_status()


def test_the_library_draws_its_side_streams_from_one_pool(dev):
    """ops.side_streams: one pool per device -- the same stream objects for every caller (a process has four hardware queues;
    streams beyond the first three share one), growing on demand; the attention block's companion is not one of them."""
    from shallow_ntc_amd import ops
    a = ops.side_streams(2, dev)
    b = ops.side_streams(3, dev)
    assert len(a) == 2 and len(b) == 3 and a[0] is b[0] and a[1] is b[1]
    assert len({s.cuda_stream for s in b}) == 3 and all(s.cuda_stream != torch.cuda.current_stream().cuda_stream for s in b)
    five = ops.side_streams(5, dev)
    assert five[:3] == b and len({s.cuda_stream for s in five}) == 5
    assert ops.side_streams(0, dev) == []
    comp = ops.companion_stream()
    assert comp is not None and comp.cuda_stream not in {s.cuda_stream for s in five}
    assert ops.companion_stream() is comp                       # one per current stream
    with torch.cuda.stream(b[0]):
        other = ops.companion_stream()
    assert other is not comp and other.cuda_stream not in {s.cuda_stream for s in five}

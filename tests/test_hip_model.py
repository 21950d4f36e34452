"""GPU parity of whole transforms and of the Model forward against the float64 oracle (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import model_np
from oracle import transforms_np as T

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def randomize(weights, rng):
    """Move biases / GDN parameters off their framework defaults so every term is exercised."""
    out = {}
    for k, v in weights.items():
        leaf = k.rsplit("/", 1)[-1]
        if leaf == "bias":
            v = (0.1 * rng.standard_normal(v.shape)).astype(np.float32)
        elif leaf == "beta":
            v = (1.0 + 0.5 * rng.random(v.shape)).astype(np.float32)
        elif leaf == "gamma":
            v = (v + 0.01 * rng.random(v.shape)).astype(np.float32)
        out[k] = v
    return out


TRANSFORMS = [  # cls, kwargs, input shape (n,h,w,c)
    ("ElicAnalysis", dict(channels=(32, 32, 64, 64)), (1, 64, 96, 3)),
    ("ElicAnalysis", dict(channels=(32, 64, 64), kernel_sizes=(5, 5, 5), strides=(2, 2, 2)), (1, 40, 24, 3)),
    ("ElicSynthesis", dict(channels=(64, 32, 32, 3)), (1, 3, 4, 64)),
    ("CNNAnalysis", dict(channels_base=32, output_channels=64), (2, 48, 32, 3)),
    ("CNNAnalysis", dict(channels_base=32, output_channels=64, activation_type="gdn"), (1, 32, 32, 3)),
    ("CNNSynthesis", dict(channels_base=32, output_channels=3), (1, 3, 2, 64)),
    ("CNNSynthesis", dict(channels_base=32, output_channels=3, activation_type="igdn"), (1, 2, 2, 64)),
    ("HyperAnalysis", dict(bottleneck_size=64), (2, 12, 8, 64)),
    ("HyperSynthesis", dict(bottleneck_size=64), (2, 3, 2, 64)),
    ("BLS2017Analysis", dict(num_filters=64), (1, 64, 48, 3)),
    ("BLS2017Synthesis", dict(num_filters=64), (1, 4, 3, 64)),
    ("MBT2018Analysis", dict(channels_base=32, output_channels=64), (1, 64, 64, 3)),
    ("MBT2018Synthesis", dict(channels_base=32, output_channels=3), (1, 4, 4, 64)),
    ("MBT2018Analysis", dict(channels_base=32, output_channels=64, gdn_alpha=2, gdn_epsilon=0.5), (1, 32, 32, 3)),
    ("MBT2018Synthesis", dict(channels_base=32, output_channels=3, gdn_alpha=2, gdn_epsilon=0.5), (1, 2, 2, 64)),
    ("HyperAnalysisSmall", dict(bottleneck_size=32), (1, 8, 8, 32)),
    ("HyperSynthesisSmall", dict(bottleneck_size=32), (1, 4, 4, 32)),
    ("JPEGLikeSynthesis", dict(kernel_size=18, strides=16), (2, 3, 4, 64)),
    ("JPEGLikeSynthesis", dict(kernel_size=16, strides=16), (1, 2, 2, 64)),
    ("JPEGLikeHyperSynthesis", dict(bottleneck_size=32, kernel_size=6), (1, 3, 3, 32)),
    ("TwoLayerSynthesis", dict(channels=(24, 3)), (1, 3, 4, 64)),
    ("TwoLayerSynthesis", dict(channels=(48, 3)), (1, 2, 3, 64)),
    ("TwoLayerResSynthesis", dict(channels=(12, 3)), (2, 4, 3, 64)),
    # registered shapes no reference config uses (common/transforms.py:291-293, 339-348; other hidden widths / output layers)
    ("JPEGLikeSynthesis", dict(kernel_size=18, strides=16, use_offset=True), (2, 3, 4, 64)),
    ("TwoLayerResSynthesis", dict(channels=(12, 3), res_type="d2s"), (2, 4, 3, 64)),
    ("TwoLayerResSynthesis", dict(channels=(8, 3), res_type="d2s", activation_type=None), (1, 2, 2, 64)),
    ("TwoLayerSynthesis", dict(channels=(16, 3), activation_type="relu"), (1, 3, 4, 64)),
    ("TwoLayerResSynthesis", dict(channels=(20, 3)), (1, 2, 3, 64)),
    ("TwoLayerSynthesis", dict(channels=(24, 3), kernel_sizes=(13, 3), activation_type="leaky_relu"), (1, 2, 2, 64)),
]


@pytest.mark.parametrize("cls,kwargs,shape", TRANSFORMS, ids=[f"{c}-{i}" for i, (c, _, _) in enumerate(TRANSFORMS)])
def test_transform_parity(cls, kwargs, shape, dev):
    from shallow_ntc_amd.common.transforms import class_builder
    rng = np.random.default_rng(len(cls) * 31 + len(kwargs))
    t = class_builder.build(cls, **kwargs)
    cin = shape[-1]
    okw = dict(kwargs)
    if cin != 3 or cls.endswith("Synthesis"):
        okw["cin"] = cin
    ref_t = T.build(cls, **okw)
    # the two independently written inventories agree on names and shapes
    assert dict(t.param_shapes(cin)) == dict(ref_t.param_shapes())
    t.build(cin, dev)
    w = randomize(t.get_weights(), rng)
    t.set_weights(w)
    x = rng.standard_normal(shape).astype(np.float32)
    got = t(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = ref_t(w, x)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 5e-5
    if hasattr(t, "forward_pixels"):                     # the decoder form of the two-layer syntheses: same image, as uint8, cropped
        from shallow_ntc_amd import ops
        xd = torch.from_numpy(x).to(dev)
        hh, ww = got.shape[1] - 3, got.shape[2] - 5
        px, sse = t.forward_pixels(xd, hh, ww)
        assert sse is None and torch.equal(px, ops.to_pixels(t(xd), hh, ww))
        refimg = torch.rand((got.shape[0], hh, ww, got.shape[3]), device=dev) - 0.5
        px2, sse2 = t.forward_pixels(xd, hh, ww, reference=refimg)
        want_sse, want_px = ops.pixels_sse(refimg, t(xd), want_pixels=True)
        assert torch.equal(px2, want_px) and torch.equal(sse2.to(torch.int64), want_sse.to(torch.int64))


def _small_cfg(synth):
    return dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 64)), synthesis=synth)


@pytest.mark.parametrize("synth", [
    dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn", res_type="conv"),
    dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16),
    dict(cls="TwoLayerSynthesis", channels=(24, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn"),
    dict(cls="TwoLayerResSynthesis", channels=(12, 3), res_type="d2s"),                       # reference transforms.py:339-348
    dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16, use_offset=True),               # :291-293
    dict(cls="TwoLayerSynthesis", channels=(16, 3), activation_type="relu"),                  # a hidden width off the fused tail
], ids=["two_layer_res", "jpeg_like", "two_layer", "two_layer_res_d2s", "jpeg_like_offset", "two_layer_16"])
def test_mshyper_model_parity(synth, dev):
    """bpp / PSNR of the HIP path vs the oracle on the same image + weights.  The latents are compared
    first; rate and distortion are then checked with the oracle evaluated on the HIP latents so that a
    float32-vs-float64 rounding flip of round(y - mu) (counted below) cannot hide an arithmetic error."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper.models import Model
    rng = np.random.default_rng(0)
    tc = _small_cfg(synth)
    model = Model(rd_lambda=0.02, transform_config=tc, device=dev)
    assert model.downsample_factor == 64
    w = randomize(model.get_weights(), rng)
    # realistic spread of scales: bias of the sigma half of the hyper-synthesis output
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[64:] = rng.uniform(-1.0, 3.0, size=64)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    model.set_weights(w)
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 100, 150, seed=3))      # pads to 128 x 192
    ref_model = model_np.Model(tc, rd_lambda=0.02)
    lat = model.infer_latent_rvs(x)
    z, y = lat.uq[0].loc.cpu().numpy(), lat.uq[1].loc.cpu().numpy()
    rz, ry = ref_model.infer_latents(w, x)
    assert rel_err(y, ry) < 5e-5 and rel_err(z, rz) < 5e-5
    # oracle on the HIP latents (float32 -> float64): isolates the generative path
    ref = ref_model.frame_loss(w, x, (z, y))
    r = model._rate_and_reconstruction(lat, want_symbols=True)
    sym = r["symbols"].cpu().numpy()
    flip = sym != ref["symbols_y"]
    assert int(flip.sum()) <= 2, int(flip.sum())
    np.testing.assert_array_equal(r["z_hat"].cpu().numpy(), ref["z_hat"])
    per_image = model.evaluate_batched(x)
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    # Rounding and arithmetic are checked separately, each without an escape: (1) the integer symbols differ from the
    # float64 oracle's only where the oracle's own y - mu sits within 1e-4 of a rounding boundary (at most 2 of 24,576
    # here: one such symbol is 1-3 bits = 1e-4 bpp on an image this small; the BASELINE tolerance on Kodak-size images
    # is asserted end to end, unconditionally, in test_hip_e2e_parity.py); (2) GIVEN the same integers, rate and
    # distortion meet the BASELINE tolerance.
    ref_s = ref_model.frame_loss(w, x, (z, y), force_symbols=sym)
    assert (ref_s["tie_distance"][flip] < 1e-4).all()
    assert abs(m["bpp"] - ref_s["bpp"]) <= 1e-4                      # BASELINE tolerance
    assert abs(m["psnr"] - ref_s["psnr"]) <= 1e-3
    assert abs(m["rd_loss"] - (m["bpp"] + 0.02 * m["mse"])) < 1e-5   # results/readme.md identity
    assert abs(np.mean([d["psnr"] for d in per_image]) - m["psnr"]) < 1e-4
    # eval JSON schema (common/eval_lib.py:92-102 rows carry msssim): 100 x 150 < 160 -> single-scale SSIM
    from oracle import ops_np as O
    q, qdb = O.image_quality(O.floats_to_pixels(x, False).astype(np.float64), ref["recon_pixels"].astype(np.float64))
    assert abs(m["msssim"] - q.mean()) < 5e-4 and abs(m["msssim_db"] - qdb.mean()) < 5e-2
    # end to end from pixels against the oracle's own float64 latents: the same two statements
    ref_e2e = ref_model.end_to_end(w, x)
    flip = sym != ref_e2e["symbols_y"]
    assert int(flip.sum()) <= 4, int(flip.sum())
    ref_e2e_s = ref_model.frame_loss(w, x, (rz, ry), force_symbols=sym)
    assert (ref_e2e_s["tie_distance"][flip] < 1e-3).all()
    assert abs(m["bpp"] - ref_e2e_s["bpp"]) <= 1e-4 and abs(m["psnr"] - ref_e2e_s["psnr"]) <= 1e-3
    # evaluate() yields one Metrics per image with the reference's scalar keys; images kept in flight on several streams
    # give exactly the numbers of the strictly serial loop, in the same order
    ms = list(model.evaluate(x))
    assert len(ms) == 2 and {"rd_loss", "bpp", "mse", "psnr", "scheduled_lr", "sched_rd_lambda"} <= set(ms[0].scalars)
    many = [x[i % 2:i % 2 + 1] for i in range(7)]
    serial = [mm.scalars_float for mm in model.evaluate(many, lookahead=1)]
    piped = [mm.scalars_float for mm in model.evaluate(many, lookahead=3)]
    assert serial == piped and serial[0] == ms[0].scalars_float and serial[1] == ms[1].scalars_float
    # images of the same shape among the next few are launched together: still each image's own numbers, in input order,
    # whatever the grouping; the recorded reconstructions are each image's own
    other = data_lib.normalize_image(data_lib.synthetic_images(2, 150, 100, seed=8))
    solo = [mm.scalars_float for mm in model.evaluate(other, lookahead=1)]
    mixed = [x[0:1], other[0:1], x[1:2], x[0:1], other[1:2], x[1:2], x[0:1]]
    want = [serial[0], solo[0], serial[1], serial[0], solo[1], serial[1], serial[0]]
    for look, group in ((2, 2), (4, 4), (3, 1), (3, 8)):
        out = list(model.evaluate(iter(mixed), lookahead=look, group=group))
        assert [mm.scalars_float for mm in out] == want, (look, group)
        rec = [mm.images["reconstruction"] for mm in out]
        assert all(r.shape[0] == 1 for r in rec) and rec[1].shape[1] > rec[1].shape[2] and rec[0].shape[1] < rec[0].shape[2]
        assert torch.equal(rec[0], rec[3]) and torch.equal(rec[2], rec[5]) and not torch.equal(rec[0], rec[2])
    # codec regions: decode(encode(x)) reproduces the evaluated reconstruction bit for bit
    z_hat, sym, bz, by = model.encode(x)
    px, sse = model.decode(z_hat, sym, (100, 150), reference=torch.from_numpy(x).to(dev))
    # uint8 pixels against the oracle's reconstruction OF THE SAME INTEGERS: identical except where the float32
    # reconstruction sits within rounding error of a .5 tie of the float64 one (never more than one code value apart)
    d = np.abs(px.cpu().numpy().astype(np.int32) - ref_s["recon_pixels"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-4, (d.max(), (d != 0).mean())
    mse_dec = sse.cpu().numpy() / (100 * 150 * 3.0)
    assert abs(mse_dec.mean() - m["mse"]) < 1e-3


def test_factorized_model_parity(dev):
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.factorized.models import Model
    rng = np.random.default_rng(1)
    tc = dict(analysis=dict(cls="BLS2017Analysis", num_filters=64), synthesis=dict(cls="BLS2017Synthesis", num_filters=64))
    model = Model(rd_lambda=0.01, transform_config=tc, device=dev)
    assert model.downsample_factor == 16
    w = randomize(model.get_weights(), rng)
    model.set_weights(w)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 70, 90, seed=5))        # pads to 80 x 96
    ref_model = model_np.Model(tc, rd_lambda=0.01, factorized=True)
    lat = model.infer_latent_rvs(x)
    y = lat.uq[0].loc.cpu().numpy()
    (ry,) = ref_model.infer_latents(w, x)
    assert rel_err(y, ry) < 5e-5
    ref = ref_model.frame_loss(w, x, (y,))
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    assert abs(m["bpp"] - ref["bpp"]) <= 1e-4 and abs(m["psnr"] - ref["psnr"]) <= 1e-3
    # lambda warm-up (mshyper/models.py:168-184): rd_lambda <= 0.01 is scaled x10 at step 0
    assert abs(m["sched_rd_lambda"] - 0.1) < 1e-9


def test_eval_workdir_from_checkpoint(tmp_path, dev):
    """reference eval.py flow: workdir/config.json + train/checkpoints/ckpt-N (TensorBundle) -> Model ->
    evaluate -> per-image results JSON; the numbers equal evaluating the in-memory model."""
    import json
    from shallow_ntc_amd.common import data_lib, eval_lib
    from shallow_ntc_amd.mshyper.models import Model
    from test_tf_checkpoint import _reference_like_checkpoint
    tc = _small_cfg(dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                         activation_type="igdn", res_type="conv"))
    cfg = dict(rd_lambda=0.02, transform_config=tc)
    model = Model(device=dev, **cfg)
    w = randomize(model.get_weights(), np.random.default_rng(4))
    model.set_weights(w)
    workdir = tmp_path / "4242" / "wid=3-mshyper-rd_lambda=0.02-bottleneck_size=64"
    ckdir = workdir / "train" / "checkpoints"
    ckdir.mkdir(parents=True)
    (workdir / "config.json").write_text(json.dumps(dict(model_config=cfg)))
    prefix = _reference_like_checkpoint(ckdir, w)
    (ckdir / "checkpoint").write_text(f'model_checkpoint_path: "{prefix.name}"\n')
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 64, 128, seed=6))
    out = eval_lib.eval_workdir(workdir, x, tmp_path / "results", device=dev)
    assert out.name == "mshyper-rd_lambda=0.02-bottleneck_size=64-step=  3-xid=4242.json"
    rows = json.loads(out.read_text())
    model._step = 3
    want = [m.scalars_float for m in model.evaluate(x)]
    assert len(rows) == 2 and rows[1]["instance_id"] == 1 and rows[0]["rd_lambda"] == 0.02 and rows[0]["bottleneck_size"] == 64
    for r, wnt in zip(rows, want):
        for k in ("bpp", "psnr", "mse", "msssim"):
            assert abs(r[k] - wnt[k]) <= 1e-6 * max(1.0, abs(wnt[k])), k


def test_profile_mode_reports_transform_times(dev):
    """Model(profile=True) adds the reference's *_time scalars (mshyper/models.py:219-224,269-271,293-295,350-351)."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper.models import Model
    cfg = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)),
               synthesis=dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16))
    model = Model(device=dev, transform_config=cfg, profile=True, quality_metrics=False)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 64, 64, seed=2))
    s = model.validation_step(x).scalars_float
    for k in ("analysis_time", "hyper_analysis_time", "hyper_synthesis_time", "synthesis_time"):
        assert 0.0 < s[k] < 1.0, (k, s[k])
    plain = Model(device=dev, transform_config=cfg, quality_metrics=False).validation_step(x).scalars_float
    assert not any(k.endswith("_time") for k in plain)
    assert plain["bpp"] == s["bpp"] and plain["psnr"] == s["psnr"]


def test_smoke_entry_point_runs(dev):
    """__graft_entry__.smoke() is what the driver runs first on the GPU box: keep it under test (128 x 192 image: the
    MS-SSIM pyramid does not fit, the metric is skipped rather than raised)."""
    import __graft_entry__ as graft
    graft.smoke()


@pytest.mark.parametrize("hw", [(128, 192), (512, 768)])
def test_attention_branches_on_two_streams_do_not_change_a_bit(hw, dev):
    """SimpleAttention (reference common/elic.py:85-100) runs trunk and branch on two streams where the launches are small
    (ops.CONCURRENT_BRANCHES: one image at 1/4 and 1/16 resolution).  Same kernels, same bits as the one-stream order, and no
    stream-K hand-off of the two launches in flight at once ever times out -- over many repetitions."""
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common import transforms as T
    t = T.class_builder.build("ElicAnalysis", channels=(192, 192, 192, 320))
    t.build(3, dev)
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.uniform(-0.5, 0.5, (1,) + hw + (3,)).astype(np.float32)).to(dev)
    assert ops.CONCURRENT_BRANCHES
    ops.CONCURRENT_BRANCHES = False
    try:
        want = t(x)
    finally:
        ops.CONCURRENT_BRANCHES = True
    ops.check_conv_status()
    for _ in range(25):
        assert torch.equal(t(x), want)
    ops.check_conv_status()                    # raises if a hand-off timed out (and would switch stream-K off for the process)

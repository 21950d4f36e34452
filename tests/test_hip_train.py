"""Training step on the GPU against float64 PyTorch autograd of the oracle loss (oracle/train_ref.py)."""
import math

import numpy as np
import pytest
import torch

import __graft_entry__ as graft

graft.load_package()
pytestmark = pytest.mark.gpu

from oracle import train_ref  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


WGRAD_CASES = [  # kind, k, s, cin, cout, n, h, w
    ("conv", 5, 2, 3, 64, 2, 32, 48),        # first analysis layer: 3 input channels (scalar gather path)
    ("conv", 3, 1, 32, 32, 2, 16, 24),
    ("conv", 1, 1, 96, 48, 1, 20, 12),
    ("conv", 5, 2, 64, 96, 2, 16, 16),
    ("conv", 5, 2, 40, 40, 1, 8, 8),
    ("convT", 5, 2, 32, 48, 2, 6, 10),
    ("convT", 3, 1, 48, 64, 1, 12, 8),
    ("convT", 13, 8, 32, 24, 2, 4, 6),       # two-layer synthesis up-conv: 169 taps, 24 channels on the gathered side
    ("convT", 5, 2, 12, 3, 1, 16, 24),       # output layer: 3 channels on the gathered side
    ("convT", 18, 16, 32, 3, 1, 3, 4),       # JPEG-like
]


@pytest.mark.parametrize("kind,k,s,cin,cout,n,h,w", WGRAD_CASES)
def test_conv_wgrad_and_bias_grad(kind, k, s, cin, cout, n, h, w, dev):
    """dW, db of sum(conv(x) * g) from sntc_conv_wgrad / sntc_bias_grad == autograd of the float64 restatement."""
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(k * 100 + s * 10 + cin)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wshape = (k, k, cin, cout) if kind == "conv" else (k, k, cout, cin)
    wt = torch.tensor(rng.standard_normal(wshape) * 0.1, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    xt = train_ref.as_input(x)
    y = train_ref.conv2d(xt, wt, bt, s) if kind == "conv" else train_ref.conv2d_transpose(xt, wt, bt, s)
    g = rng.standard_normal((n, y.shape[2], y.shape[3], cout)).astype(np.float32)
    (y * train_ref.as_input(g)).sum().backward()
    dw = torch.full(wshape, 7.0, dtype=torch.float32, device=dev)              # must be overwritten
    db = torch.empty((cout,), dtype=torch.float32, device=dev)
    xd, gd = torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev)
    ops.conv_wgrad(kind, k, s, cin, cout, xd, gd, dw)
    ops.bias_grad(gd, db)
    assert _rel(dw.cpu().numpy(), wt.grad.numpy()) < 2e-5
    assert _rel(db.cpu().numpy(), bt.grad.numpy()) < 2e-5
    ops.conv_wgrad(kind, k, s, cin, cout, xd, gd, dw, accumulate=True)           # accumulate: exactly twice
    assert _rel(dw.cpu().numpy(), 2 * wt.grad.numpy()) < 2e-5
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(kind, k, s, cin, cout, xd, gd, dw2)                           # deterministic: bit-identical on a re-run
    ops.conv_wgrad(kind, k, s, cin, cout, xd, gd, dw)
    assert torch.equal(dw, dw2)


def test_elementwise_training_kernels(dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(3)
    g = rng.standard_normal((2, 5, 7, 8)).astype(np.float32)
    y = rng.standard_normal((2, 5, 7, 8)).astype(np.float32)
    gd, yd = torch.from_numpy(g).to(dev), torch.from_numpy(y).to(dev)
    np.testing.assert_array_equal(ops.act_backward(gd, yd, "relu").cpu().numpy(), g * (y > 0))
    np.testing.assert_array_equal(ops.act_backward(gd, yd, "leaky_relu").cpu().numpy(), g * np.where(y > 0, 1.0, 0.2).astype(np.float32))
    sg = 1.0 / (1.0 + np.exp(-y))
    np.testing.assert_allclose(ops.act_backward(gd, torch.from_numpy(sg).to(dev), "sigmoid").cpu().numpy(), g * sg * (1 - sg), rtol=1e-6)
    t = rng.standard_normal(g.shape).astype(np.float32)
    td, sd = torch.from_numpy(t).to(dev), torch.from_numpy(sg).to(dev)
    np.testing.assert_allclose(ops.gate_forward(yd, td, sd).cpu().numpy(), y + t * sg, rtol=1e-6, atol=1e-7)
    g_t, g_s = ops.gate_backward(gd, td, sd)
    np.testing.assert_allclose(g_t.cpu().numpy(), g * sg, rtol=1e-6)
    np.testing.assert_allclose(g_s.cpu().numpy(), g * t * sg * (1 - sg), rtol=1e-5, atol=1e-7)
    a = gd.clone()
    ops.axpy(a, yd, 0.5)
    np.testing.assert_allclose(a.cpu().numpy(), g + 0.5 * y, rtol=1e-6)
    assert abs(float(ops.sumsq(gd).item()) - float((g.astype(np.float64) ** 2).sum())) < 1e-6 * g.size
    u = ops.noise_add(torch.zeros((4096,), device=dev), None, seed=5, step=9).cpu().numpy()
    assert -0.5 < u.min() and u.max() < 0.5 and abs(u.mean()) < 0.02 and abs(u.var() - 1 / 12) < 0.005
    u2 = ops.noise_add(torch.zeros((4096,), device=dev), None, seed=5, step=9).cpu().numpy()
    u3 = ops.noise_add(torch.zeros((4096,), device=dev), None, seed=5, step=10).cpu().numpy()
    assert np.array_equal(u, u2) and not np.array_equal(u, u3)


def test_rate_gradient_through_a_saturated_scale_index(dev):
    """exp(raw) > 63 saturates the scale index; tfc bounds it with `identity_if_towards`, so such an element still gets
    d bits / d raw when descent would pull it back (here: whenever sigma = 256 is too wide for v) and none otherwise."""
    from oracle import train_ref
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(7)
    c = 8
    mu = rng.standard_normal((2, 4, 5, c)).astype(np.float32)
    raw = rng.uniform(-2.0, 5.0, size=mu.shape).astype(np.float32)          # ln 63 = 4.14: about an eighth saturate
    raw[0, 0, 0, :4] = [4.2, 4.5, 4.9, 4.1431]
    v = rng.laplace(0, 3, size=mu.shape).astype(np.float32)
    v[0, 0, 0, 0], v[0, 0, 0, 1] = 900.0, 0.25                            # sigma = 256 too narrow / too wide
    hyper = np.concatenate([mu, raw], -1)
    bits, dv, dr = ops.noisy_normal(torch.from_numpy(v + mu).to(dev), torch.from_numpy(hyper).to(dev))
    vt = torch.from_numpy(v.astype(np.float64)).requires_grad_(True)
    rt = torch.from_numpy(raw.astype(np.float64)).requires_grad_(True)
    ref = train_ref.noisy_normal_bits(vt, rt)
    ref.sum().backward()
    np.testing.assert_allclose(bits.cpu().numpy(), ref.detach().numpy().sum(axis=(1, 2, 3)), rtol=2e-5)
    # float32 derivatives of far-tail terms (|v| / sigma up to ~60 at sigma = 0.11: gradients of +-900 bits per unit)
    np.testing.assert_allclose(dv.cpu().numpy(), vt.grad.numpy(), rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(dr.cpu().numpy(), rt.grad.numpy(), rtol=1e-3, atol=2e-6)
    sat = np.exp(raw.astype(np.float64)) > 63
    got = dr.cpu().numpy()
    assert sat.sum() >= 8 and (got[sat] >= 0).all() and (got[sat] > 0).any() and (got[sat] == 0).any()
    assert got[0, 0, 0, 0] == 0.0 and got[0, 0, 0, 1] > 0.0


def _small_model(dev, synthesis, analysis=None, uq="unoise"):
    from shallow_ntc_amd.mshyper.models import Model
    cfg = dict(analysis=analysis or dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)), synthesis=synthesis)
    model = Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=1000, latent_config=dict(uq=dict(method=uq)),
                  optimizer_config=dict(learning_rate=1e-3, global_clipnorm=1.0, warmup_steps=0), quality_metrics=False)
    w = dict(model.get_weights())
    rng = np.random.default_rng(11)
    for k, v in w.items():                       # biases / gate inputs away from zero so every path carries gradient
        if k.endswith("/bias"):
            w[k] = (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
    w["hyper_analysis/layer_2/bias"] = (1.5 * rng.standard_normal(w["hyper_analysis/layer_2/bias"].shape)).astype(np.float32)  # z away from 0
    for k in ("prior/factor_0", "prior/factor_1"):
        if k in w:
            w[k] = (0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
    model.set_weights(w)
    return model, cfg


SYNTHESES = [
    dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn", res_type="conv"),
    dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16),
    dict(cls="TwoLayerSynthesis", channels=(24, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn"),
]
CNN = dict(cls="CNNAnalysis", channels_base=32, output_channels=32)      # two_layer_syn2.py: CNN analysis, leaky_relu, mixedq


@pytest.mark.parametrize("synthesis,analysis,uq", [(SYNTHESES[0], None, "unoise"), (SYNTHESES[1], None, "unoise"),
                                                   (SYNTHESES[2], CNN, "mixedq")],
                         ids=["two_layer_res", "jpeg_like", "cnn_two_layer_mixedq"])
def test_loss_and_gradients_match_autograd(synthesis, analysis, uq, dev):
    """Every d loss / d variable of the HIP backward pass against float64 autograd of the oracle training loss under the
    same uniform noise: ELIC analysis (residual blocks, attention gates), hyper transforms, entropy models incl. the
    deep-factorized prior variables, two-layer synthesis incl. the reparameterised IGDN variables."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.train import GDN_BETA_MIN, Trainer, gdn_raw
    model, cfg = _small_model(dev, synthesis, analysis, uq)
    tr = Trainer(model, seed=1)
    n, h, w = 2, 64, 128
    x = data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=5))
    rng = np.random.default_rng(7)
    nz = rng.uniform(-0.5, 0.5, size=(n, h // 64, w // 64, 32)).astype(np.float32)
    ny = rng.uniform(-0.5, 0.5, size=(n, h // 16, w // 16, 32)).astype(np.float32)
    lam = 0.02
    xd = torch.from_numpy(x).to(dev)
    out = tr.loss_and_grads(xd, lam, torch.from_numpy(nz).to(dev), torch.from_numpy(ny).to(dev))
    got = tr.store.export(tr.store.grad)
    params = dict(model.get_weights())
    raw_names = ()
    if "synthesis/act/beta" in params:
        params["synthesis/act/beta"] = (gdn_raw(params["synthesis/act/beta"], GDN_BETA_MIN), GDN_BETA_MIN)
        params["synthesis/act/gamma"] = (gdn_raw(params["synthesis/act/gamma"], 0.0), 0.0)
        raw_names = ("synthesis/act/beta", "synthesis/act/gamma")
    ref = train_ref.loss_and_grads(cfg, params, x, nz, ny, lam, gdn_raw_names=raw_names, uq=uq)
    bits_z, bits_y = out["bits_z"].cpu().numpy(), out["bits_y"].cpu().numpy()
    assert _rel(bits_z, ref["bits_z"]) < 2e-5 and _rel(bits_y, ref["bits_y"]) < 2e-5
    assert _rel(out["recon"].cpu().numpy(), ref["recon"]) < 5e-5
    mse = float((out["sse"].cpu().numpy() / (h * w * 3)).mean())
    assert abs(mse - ref["mse"]) < 2e-5 * ref["mse"]
    rg = dict(ref["grads"])
    if "synthesis/base_conv/kernel" in rg:                                       # the store keeps [base | res] as one kernel
        rg["synthesis/up/kernel"] = np.concatenate([rg.pop("synthesis/base_conv/kernel"), rg.pop("synthesis/res/kernel")], axis=2)
        rg["synthesis/up/bias"] = np.concatenate([rg.pop("synthesis/base_conv/bias"), rg.pop("synthesis/res/bias")])
    elif "synthesis/conv1/kernel" in rg:
        rg["synthesis/up/kernel"], rg["synthesis/up/bias"] = rg.pop("synthesis/conv1/kernel"), rg.pop("synthesis/conv1/bias")
    if "synthesis/act/beta" in rg:
        rg["synthesis/act/beta_raw"] = rg.pop("synthesis/act/beta")
        rg["synthesis/act/gamma_raw"] = rg.pop("synthesis/act/gamma")
    assert set(rg) == set(got)
    worst = max(((_rel(got[k], rg[k]), k) for k in rg), key=lambda t: t[0])
    for k in rg:
        assert np.abs(rg[k]).max() > 0, f"{k}: reference gradient is identically zero (test would be vacuous)"
    assert worst[0] < 2e-5, worst          # measured: 3e-6
    # the weight / bias gradients run on a side stream next to the input-gradient chain (Trainer.overlap_wgrad): the same
    # kernels in another order of launches -- every gradient bit for bit, run after run
    # (the deep-factorized prior's own gradients are float atomics: equal to ~1e-7 only, overlap or not)
    assert tr.overlap_wgrad
    for overlap in (False, True, True):
        tr.overlap_wgrad = overlap
        tr.store.grad.zero_()
        tr.loss_and_grads(xd, lam, torch.from_numpy(nz).to(dev), torch.from_numpy(ny).to(dev))
        again = tr.store.export(tr.store.grad)
        for k in got:
            if k.startswith("prior/"):
                assert _rel(again[k], got[k]) < 1e-6, (k, overlap)
            else:
                assert np.array_equal(again[k], got[k]), (k, overlap)


@pytest.mark.parametrize("which", ["mbt2018", "bls2017"])
def test_gdn_and_signal_conv_stacks_train(which, dev):
    """BASELINE configs 1 and 2 under Model.train_step (reference mshyper/models.py:375-383 applies to every config):
    tfc.SignalConv2D layers train their real-DFT coefficients, tfc.GDN / GDN1 their reparameterised beta / gamma.  Every
    d loss / d variable against float64 autograd of the oracle loss (kernel gradients pulled back with M^T, GDN gradients
    through the reparameterisation), for the mean-scale model with MBT2018 transforms and for the factorized-prior model
    with BLS2017 transforms; then a few optimizer steps lower the loss and the exported weights evaluate."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.common.tf_checkpoint import irdft_matrix
    from shallow_ntc_amd.train import GDN_BETA_MIN, Trainer, gdn_raw
    common = dict(device=dev, rd_lambda=0.02, scheduled_num_steps=1000, latent_config=dict(uq=dict(method="unoise")),
                  optimizer_config=dict(learning_rate=1e-3, global_clipnorm=1.0, warmup_steps=0), quality_metrics=False)
    if which == "mbt2018":
        from shallow_ntc_amd.mshyper.models import Model
        cfg = dict(analysis=dict(cls="MBT2018Analysis", channels_base=32, output_channels=32),
                   synthesis=dict(cls="MBT2018Synthesis", channels_base=32))
        factorized = False
    else:
        from shallow_ntc_amd.factorized.models import Model
        cfg = dict(analysis=dict(cls="BLS2017Analysis", num_filters=32), synthesis=dict(cls="BLS2017Synthesis", num_filters=32))
        factorized = True
    model = Model(transform_config=cfg, **common)
    w = dict(model.get_weights())
    rng = np.random.default_rng(3)
    for k, v in w.items():
        if k.endswith("/bias"):
            w[k] = (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
        elif k.endswith("/beta"):
            w[k] = (1.0 + 0.5 * rng.random(v.shape)).astype(np.float32)
        elif k.endswith("/gamma"):
            w[k] = (0.1 * np.eye(v.shape[0]) + 0.02 * rng.random(v.shape)).astype(np.float32)
        elif k.startswith("prior/factor"):
            w[k] = (0.3 * rng.standard_normal(v.shape)).astype(np.float32)
    if not factorized:
        w["hyper_analysis/layer_2/bias"] = (1.5 * rng.standard_normal(w["hyper_analysis/layer_2/bias"].shape)).astype(np.float32)
    model.set_weights(w)
    tr = Trainer(model, seed=1)
    n, h, wd = 2, 64, 128
    x = data_lib.normalize_image(data_lib.synthetic_images(n, h, wd, seed=5))
    nz = rng.uniform(-0.5, 0.5, size=(n, h // 64, wd // 64, 32)).astype(np.float32)
    ny = rng.uniform(-0.5, 0.5, size=(n, h // 16, wd // 16, 32)).astype(np.float32)
    lam = 0.02
    xd = torch.from_numpy(x).to(dev)
    out = tr.loss_and_grads(xd, lam, None if factorized else torch.from_numpy(nz).to(dev), torch.from_numpy(ny).to(dev))
    got = tr.store.export(tr.store.grad)
    params, raw_names = dict(w), []
    for k in w:
        if k.endswith("/beta") or k.endswith("/gamma"):
            mn = GDN_BETA_MIN if k.endswith("/beta") else 0.0
            params[k] = (gdn_raw(w[k], mn), mn)
            raw_names.append(k)
    ref = train_ref.loss_and_grads(cfg, params, x, nz, ny, lam, gdn_raw_names=tuple(raw_names), factorized=factorized)
    assert _rel(out["bits_y"].cpu().numpy(), ref["bits_y"]) < 2e-5 and _rel(out["recon"].cpu().numpy(), ref["recon"]) < 5e-5
    rg = {}
    for k, g in ref["grads"].items():
        if k in raw_names:
            rg[k + "_raw"] = g
        elif k.endswith("/kernel") and k.split("/")[0] in ("analysis", "synthesis"):      # SignalConv2D: d rdft = M^T d kernel
            kh, kw = g.shape[:2]
            rg[k[:-len("kernel")] + "rdft"] = irdft_matrix((kh, kw)).T @ g.reshape(kh * kw, -1)
        else:
            rg[k] = g
    assert set(rg) == set(got), set(rg) ^ set(got)
    for k in rg:
        assert np.abs(rg[k]).max() > 0, f"{k}: reference gradient is identically zero (test would be vacuous)"
    worst = max(((_rel(got[k], rg[k]), k) for k in rg), key=lambda t: t[0])
    assert worst[0] < 3e-5, worst
    # a few real steps: loss goes down, exported weights load into the inference model and evaluate
    losses = [tr.train_step(x)["rd_loss"] for _ in range(8)]
    assert losses[-1] < losses[0] and math.isfinite(losses[-1])
    tr.sync_model()
    rows = model.evaluate_batched(x)
    assert all(math.isfinite(r["bpp"]) and math.isfinite(r["psnr"]) for r in rows)
    new = model.get_weights()
    assert any(np.abs(new[k] - w[k]).max() > 0 for k in w if k.endswith("/kernel"))


def test_train_step_updates_and_export(dev):
    """One full step = clip by global norm + Keras Adam on every variable (checked against the oracle arithmetic), plans
    re-packed: the second step's loss equals a fresh model built from the exported weights; the loss goes down."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.train import Trainer
    model, cfg = _small_model(dev, SYNTHESES[0])
    tr = Trainer(model, seed=3)
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 64, 64, seed=9))
    before = tr.store.export()
    m0 = tr.train_step(x)
    g = tr.store.export(tr.store.grad)
    after = tr.store.export()
    scale, norm = train_ref.clip_scale(list(g.values()), 1.0)
    assert abs(norm - m0["grad_norm"]) < 1e-4 * norm
    for k in before:
        want, _, _ = train_ref.adam_update(before[k].astype(np.float64), scale * g[k].astype(np.float64), 0.0, 0.0, 1e-3, 1)
        assert np.abs(after[k] - want).max() < 2e-6, k
    losses = [m0["rd_loss"]] + [tr.train_step(x)["rd_loss"] for _ in range(12)]
    assert losses[-1] < losses[0], losses
    assert model._step == 13 and math.isfinite(losses[-1])
    # exported weights reproduce the trainer's forward in the inference model
    tr.sync_model()
    xd = torch.from_numpy(x).to(dev)
    lat = model.infer_latent_rvs(xd)
    y_tr, _ = tr.analysis.fwd(xd)
    assert _rel(lat.uq[1].loc.cpu().numpy(), y_tr.cpu().numpy()) < 1e-5


def test_data_parallel_gradient_is_the_mean_of_shard_gradients(dev):
    """The data-parallel contract of the training step, checked inside one process (no process is ever spawned from a
    GPU-initialised test run): every rank normalises its loss by its LOCAL batch, so the mean of the shard gradients --
    what BucketReducer.finish() hands to Adam -- equals the full-batch gradient.  The all-reduce itself is covered by
    tests/test_distributed.py (2 gloo ranks, CPU) and by tools/ddp_check.py under torchrun."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.train import Trainer
    n, h, w = 4, 64, 64
    x = data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=21))
    rng = np.random.default_rng(2)
    nz = rng.uniform(-0.5, 0.5, size=(n, 1, 1, 32)).astype(np.float32)
    ny = rng.uniform(-0.5, 0.5, size=(n, 4, 4, 32)).astype(np.float32)
    model, _ = _small_model(dev, SYNTHESES[0])
    tr = Trainer(model, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    launched = []
    tr.loss_and_grads(t(x), 0.02, t(nz), t(ny), on_bucket=launched.append)
    assert launched == list(Trainer.BUCKETS)                       # every bucket reported, in backward order
    full = tr.store.grad.clone()
    acc = torch.zeros_like(full)
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        tr.loss_and_grads(t(x[sl]), 0.02, t(nz[sl]), t(ny[sl]))
        acc += tr.store.grad
    acc *= 0.5
    scale = float(full.abs().max())
    assert float((acc - full).abs().max()) < 2e-5 * scale
    slices = tr._bucket_slices()
    assert list(slices) == list(Trainer.BUCKETS) and list(slices.values())[-1][1] == tr.store.total
    lo = 0
    for name, (a, b) in slices.items():                            # contiguous, 16-byte aligned, covering the buffer
        assert a == lo and b >= a and a % 4 == 0
        lo = b


def test_train_eval_compress_round_trip(dev, tmp_path):
    """The rows of SURVEY.md 8 working together: a small model is trained for some steps (f4), its variables go back
    into the inference model, which evaluates (a17/a18), and compresses / decompresses a real bitstream (f2) whose
    size tracks the estimated rate; training lowers the rate-distortion loss the evaluation reports."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper.models import Model
    cfg = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)),
               synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn"))
    model = Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=400,
                  optimizer_config=dict(learning_rate=2e-3, global_clipnorm=1.0, warmup_steps=0), quality_metrics=False)
    x = data_lib.normalize_image(data_lib.synthetic_images(4, 128, 128, seed=31))
    before = np.mean([m.scalars_float["rd_loss"] for m in model.evaluate(x)])
    for _ in range(60):
        m = model.train_step(x)
    assert set(("rd_loss", "bpp", "mse", "psnr", "scheduled_lr", "sched_rd_lambda")) <= set(m.scalars_float)
    model.trainer.sync_model()
    rows = model.evaluate_batched(x)
    after = np.mean([r["rd_loss"] for r in rows])
    assert after < 0.7 * before, (before, after)
    # checkpoint in the reference's TensorBundle layout -> eval_lib.load_latest_ckpt -> identical evaluation
    from shallow_ntc_amd.common import eval_lib
    model.trainer.save_checkpoint(tmp_path)
    restored = eval_lib.load_latest_ckpt(tmp_path, device=dev)
    assert restored._step == 60
    rows2 = restored.evaluate_batched(x)
    assert [r["bpp"] for r in rows2] == [r["bpp"] for r in rows] and [r["psnr"] for r in rows2] == [r["psnr"] for r in rows]
    blob = model.compress(x)
    px = model.decompress(blob)
    z_hat, sym, bits_z, bits_y = model.encode(x)
    assert torch.equal(px, model.decode(z_hat, sym, (128, 128)))
    est = float(bits_z.sum() + bits_y.sum())
    overhead = 8 * (4 * 2 * 256 + 64)                        # 2 streams x 256 B per image + header / length fields
    assert 0.97 * est < 8 * len(blob) < 1.06 * est + overhead, (8 * len(blob), est)


def test_non_finite_loss_leaves_the_training_state_untouched(dev):
    """check_numerics semantics (reference mshyper/models.py:308-309,356 fire inside the loss, before apply_gradients): a
    NaN batch raises NonFiniteError and neither the variables, the Adam moments, the packed plans nor the step move."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.train import Trainer
    model, _ = _small_model(dev, SYNTHESES[1])
    tr = Trainer(model, seed=3)
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 64, 64, seed=9))
    tr.train_step(x)
    snap = [tr.store.param.clone(), tr.store.m.clone(), tr.store.v.clone()]
    good = tr.loss_and_grads(torch.from_numpy(x).to(dev), 0.02)
    loss_before = float(good["bits_y"].sum())
    bad = x.copy()
    bad[0, 3, 4, 1] = np.nan
    with pytest.raises(capi.NonFiniteError):
        tr.train_step(bad)
    assert tr.step_count == 1 and model._step == 1
    for a, b in zip(snap, [tr.store.param, tr.store.m, tr.store.v]):
        assert torch.equal(a, b)
    assert float(tr.loss_and_grads(torch.from_numpy(x).to(dev), 0.02)["bits_y"].sum()) == loss_before     # plans not re-packed from NaNs
    assert math.isfinite(tr.train_step(x)["rd_loss"]) and tr.step_count == 2


def test_resume_continues_step_schedules_and_adam_moments(dev, tmp_path):
    """A run that is stopped after 3 steps, restored from its own workdir and continued takes the SAME 4th and 5th steps
    as the uninterrupted run: the step (learning-rate warm-up, Adam bias correction, noise stream), the variables and
    the Adam moments all come back (reference train_lib.py:123-126,190 restore_or_initialize)."""
    from shallow_ntc_amd.common import data_lib, eval_lib
    from shallow_ntc_amd.mshyper.models import Model
    from shallow_ntc_amd.train import Trainer
    cfg = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)), synthesis=dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16))
    kw = dict(rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=100, quality_metrics=False,
              optimizer_config=dict(learning_rate=1e-3, global_clipnorm=1.0, warmup_steps=10))
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 64, 64, seed=9))
    a = Model(device=dev, **kw)
    lrs = [a.train_step(x).scalars_float["scheduled_lr"] for _ in range(3)]
    assert lrs == pytest.approx([1e-4, 2e-4, 3e-4])                      # min(1, (step + 1) / 10): first update is not zero
    a.trainer.save_checkpoint(tmp_path)
    b = eval_lib.load_latest_ckpt(tmp_path, device=dev)
    assert b._step == 3
    b.trainer = Trainer(b, seed=b._seed)
    assert b.trainer.step_count == 3 and b.trainer.restore_optimizer(eval_lib.latest_checkpoint(tmp_path / "train" / "checkpoints"))
    for _ in range(2):
        ma, mb = a.train_step(x).scalars_float, b.train_step(x).scalars_float
        assert ma["scheduled_lr"] == mb["scheduled_lr"]
        assert abs(ma["rd_loss"] - mb["rd_loss"]) <= 1e-12 * ma["rd_loss"]       # per-image double sums land by atomics: last-bit order effects
    assert ma["scheduled_lr"] == pytest.approx(5e-4) and a._step == b._step == 5
    assert torch.equal(a.trainer.store.param, b.trainer.store.param) and torch.equal(a.trainer.store.m, b.trainer.store.m)
    # without the optimizer file (a reference checkpoint): fresh moments, but the step still continues
    c = eval_lib.load_latest_ckpt(tmp_path, device=dev)
    c.trainer = Trainer(c, seed=c._seed)
    assert not c.trainer.restore_optimizer(tmp_path / "nowhere" / "ckpt-3") and c.trainer.step_count == 3
    # weights set from outside drop the stale trainer state
    a.set_weights(b.get_weights())
    assert a.trainer is None


def test_train_eval_loop_and_itinf_loop_drivers(dev, tmp_path):
    """common/train_lib.simple_train_eval_loop (reference :87-258) and common/itinf_lib.itinf_on_data_batch (:26-93):
    logging / evaluation / checkpoint cadence, JSON-lines records, a restorable checkpoint, SGA variables returned."""
    import json
    from shallow_ntc_amd.common import data_lib, eval_lib, itinf_lib, train_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    cfg = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)), synthesis=dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16))
    model = Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=12,
                  optimizer_config=dict(learning_rate=1e-3, global_clipnorm=1.0, warmup_steps=0), quality_metrics=False)
    batches = [data_lib.normalize_image(data_lib.synthetic_images(2, 64, 64, seed=s)) for s in range(3)]

    def forever():
        while True:
            yield from batches

    rows = train_lib.simple_train_eval_loop(dict(num_steps=12, log_metrics_every_steps=4, checkpoint_every_steps=6, eval_every_steps=6),
                                            tmp_path, model, forever(), batches[:2])
    assert [r["step"] for r in rows] == [0, 4, 8] and rows[-1]["rd_loss"] < rows[0]["rd_loss"]
    train_rec = [json.loads(l) for l in (tmp_path / "train" / "record.jsonl").read_text().splitlines()]
    val_rec = [json.loads(l) for l in (tmp_path / "val" / "record.jsonl").read_text().splitlines()]
    assert [r["step"] for r in train_rec] == [0, 4, 8] and [r["step"] for r in val_rec] == [6, 12]
    # CheckpointManager(max_to_keep=1) (reference train_lib.py:124-126): only the newest survives, named by the state file
    assert sorted(p.name for p in (tmp_path / "train" / "checkpoints").glob("*.index")) == ["ckpt-12.index"]
    assert 'model_checkpoint_path: "ckpt-12"' in (tmp_path / "train" / "checkpoints" / "checkpoint").read_text()
    restored = eval_lib.load_latest_ckpt(tmp_path, device=dev)
    assert restored._step == 12
    # max_ckpts_to_keep > 1 (reference train_lib.py:124-126): the newest N by step stay and are all listed; files that are not
    # ckpt-<step>.* of this naming scheme (say, a reference-written side file) are never deleted
    keep_dir = tmp_path / "keep2"
    (keep_dir / "train" / "checkpoints").mkdir(parents=True)
    (keep_dir / "train" / "checkpoints" / "ckpt-notes.txt").write_text("not ours to delete")
    model2 = Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=9,
                   optimizer_config=dict(learning_rate=1e-3, global_clipnorm=1.0, warmup_steps=0), quality_metrics=False)
    train_lib.simple_train_eval_loop(dict(num_steps=9, log_metrics_every_steps=100, checkpoint_every_steps=3, max_ckpts_to_keep=2),
                                     keep_dir, model2, forever(), batches[:1])
    ck = keep_dir / "train" / "checkpoints"
    assert sorted(p.name for p in ck.glob("*.index")) == ["ckpt-6.index", "ckpt-9.index"] and (ck / "ckpt-notes.txt").exists()
    state = (ck / "checkpoint").read_text()
    assert 'model_checkpoint_path: "ckpt-9"' in state and state.count("all_model_checkpoint_paths") == 2 and not (ck / "checkpoint.tmp").exists()
    # recency, not step number (CheckpointManager): a reused workdir that holds a bundle with a HIGHER step keeps the checkpoint
    # just written and drops the stale one; the state file never names a deleted bundle
    reuse = tmp_path / "reuse"
    rck = reuse / "train" / "checkpoints"
    rck.mkdir(parents=True)
    import os
    for suffix in (".index", ".data-00000-of-00001"):
        (rck / f"ckpt-99{suffix}").write_bytes(b"stale")
        os.utime(rck / f"ckpt-99{suffix}", (1, 1))
    assert model2.trainer is not None and model2.trainer.step_count == 9
    model2.trainer.save_checkpoint(reuse)
    assert sorted(p.name for p in rck.glob("*.index")) == ["ckpt-9.index"]
    assert 'model_checkpoint_path: "ckpt-9"' in (rck / "checkpoint").read_text() and "ckpt-99" not in (rck / "checkpoint").read_text()
    assert eval_lib.load_latest_ckpt(reuse, device=dev)._step == 9
    a = restored.validation_step(batches[0]).scalars_float
    b = model.validation_step(batches[0]).scalars_float
    assert a["bpp"] == b["bpp"] and a["psnr"] == b["psnr"]
    # SGA loop on one batch with the itinf latent config (mshyper/configs/itinf.py)
    sga = Model(device=dev, rd_lambda=0.02, transform_config=cfg, quality_metrics=False, scheduled_num_steps=20,
                optimizer_config=dict(learning_rate=5e-3, warmup_steps=0), latent_config=configs.itinf()["latent_config"])
    sga.set_weights(restored.get_weights())
    tm, vm, variables = itinf_lib.itinf_on_data_batch(dict(num_steps=20, log_metrics_every_steps=5, eval_every_steps=10), None, None, sga, batches[0])
    assert [r["step"] for r in tm] == [0, 5, 10, 15] and [r["step"] for r in vm] == [10, 20]
    assert set(variables) == {"z_loc", "y_loc"} and variables["y_loc"].shape == (2, 4, 4, 32)

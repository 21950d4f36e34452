"""SGA iterative inference on the GPU against the float64 oracle (-m gpu):
the stochastic-rounding sample, the training-mode loss, its gradients w.r.t. the latents (finite
differences of the oracle loss with the Gumbel noise held fixed), Adam, and a short optimisation."""
import math

import numpy as np
import pytest
import torch

from oracle import model_np
from oracle import ops_np as O

pytestmark = pytest.mark.gpu

TC = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 64)),
          synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                         activation_type="igdn", res_type="conv"))
ITINF = dict(scheduled_num_steps=3000,
             optimizer_config=dict(learning_rate=5e-3, reduce_lr_after=0.9, reduce_lr_factor=0.1, global_clipnorm=None,
                                   warmup_until=0.0),
             latent_config=dict(uq=dict(method="sga", tau_r=5e-4, tau_ub=0.5, tau_t0=200)), offset_heuristic=False)


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def gumbel(rng, shape):
    return (-np.log(-np.log(rng.uniform(1e-6, 1 - 1e-6, size=shape + (2,))))).astype(np.float32)


def make_model(dev, synth=None, lam=0.02):
    from shallow_ntc_amd.mshyper.models import Model
    tc = dict(TC) if synth is None else dict(TC, synthesis=synth)
    model = Model(rd_lambda=lam, transform_config=tc, device=dev, **ITINF)
    rng = np.random.default_rng(5)
    w = dict(model.get_weights())
    for k in list(w):
        leaf = k.rsplit("/", 1)[-1]
        if leaf == "bias":
            w[k] = (0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif leaf == "beta":
            w[k] = (1 + 0.5 * rng.random(w[k].shape)).astype(np.float32)
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[64:] = rng.uniform(-1, 2.5, size=64)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    for k in w:
        if k.startswith("prior/"):
            w[k] = (w[k] + 0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
    model.set_weights(w)
    return model, w, tc


def test_sga_sample_and_derivative(dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(0)
    p = model_np.init_deep_factorized(8, rng, (3, 3))
    ms, bs, fs = model_np._prior_lists(p)
    prior = ops.DeepFactorizedPrior(ms, bs, fs)
    z = (3 * rng.standard_normal((2, 3, 4, 8))).astype(np.float32)
    z[0, 0, 0, :3] = [2.0, -1.0, 0.5]                      # integers (floor == ceil) and an exact tie
    g = gumbel(rng, z.shape)
    for tau in [0.5, 0.1]:
        zt, sp, db, bits = ops.sga_factorized_fwd(prior, t(z, dev), tau, noise=t(g, dev))
        ref = O.sga_round(z.astype(np.float64), tau, g.astype(np.float64))
        np.testing.assert_allclose(zt.cpu().numpy(), ref, atol=2e-5)
        h = 1e-4
        fd = (O.sga_round(z + h, tau, g.astype(np.float64)) - O.sga_round(z - h, tau, g.astype(np.float64))) / (2 * h)
        frac = np.abs(z - np.rint(z))
        ok = frac > 1e-2                                      # away from the kinks at integers
        np.testing.assert_allclose(sp.cpu().numpy()[ok], fd[ok], rtol=2e-3, atol=2e-3)
        ref_bits = O.deep_factorized_logprob(ref, ms, bs, fs).sum(axis=(1, 2, 3)) / -math.log(2)
        np.testing.assert_allclose(bits.cpu().numpy(), ref_bits, rtol=2e-5)
        lp = lambda v: O.deep_factorized_logprob(v, ms, bs, fs) / -math.log(2)
        fdb = (lp(ref + h) - lp(ref - h)) / (2 * h)
        np.testing.assert_allclose(db.cpu().numpy(), fdb, rtol=2e-3, atol=2e-4)
    # generator path: bounded between floor and ceil, different per step, reproducible per (seed, step)
    a = ops.sga_factorized_fwd(prior, t(z, dev), 0.3, seed=7, step=1)[0].cpu().numpy()
    b = ops.sga_factorized_fwd(prior, t(z, dev), 0.3, seed=7, step=1)[0].cpu().numpy()
    c = ops.sga_factorized_fwd(prior, t(z, dev), 0.3, seed=7, step=2)[0].cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert (a != c).any() and (a >= np.floor(z) - 1e-6).all() and (a <= np.ceil(z) + 1e-6).all()


def test_normal_rate_gradients(dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(1)
    c = 8
    mu = rng.standard_normal((1, 3, 3, c)).astype(np.float32)
    raw = rng.uniform(-2.5, 3.5, size=mu.shape).astype(np.float32)
    y = (mu + rng.laplace(0, 2, size=mu.shape)).astype(np.float32)
    g = gumbel(rng, mu.shape)
    tau = 0.4
    hyper = np.concatenate([mu, raw], -1)
    yt, sp, dv, dr, bits = ops.sga_normal_fwd(t(y, dev), t(hyper, dev), tau, noise=t(g, dev))
    v = O.sga_round(y.astype(np.float64) - mu, tau, g.astype(np.float64))
    np.testing.assert_allclose(yt.cpu().numpy(), v + mu, atol=3e-5)

    def bits_of(v_, raw_):
        sig = O.scale_fn(np.clip(np.exp(raw_), 0, 63))
        return O.noisy_normal_logprob(v_, sig) / -math.log(2)

    np.testing.assert_allclose(bits.cpu().numpy(), bits_of(v, raw.astype(np.float64)).sum(axis=(1, 2, 3)), rtol=2e-5)
    h = 1e-5
    fdv = (bits_of(v + h, raw.astype(np.float64)) - bits_of(v - h, raw.astype(np.float64))) / (2 * h)
    fdr = (bits_of(v, raw.astype(np.float64) + h) - bits_of(v, raw.astype(np.float64) - h)) / (2 * h)
    np.testing.assert_allclose(dv.cpu().numpy(), fdv, rtol=3e-3, atol=1e-3)
    np.testing.assert_allclose(dr.cpu().numpy(), fdr, rtol=3e-3, atol=1e-3)


def test_adam_matches_keras_formula(dev):
    from shallow_ntc_amd import ops
    rng = np.random.default_rng(2)
    p = rng.standard_normal(1000).astype(np.float32)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    pd, md, vd = t(p, dev), t(m, dev), t(v, dev)
    p64, m64, v64 = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in range(1, 4):
        g = rng.standard_normal(1000).astype(np.float32)
        ops.adam_step(pd, t(g, dev), md, vd, 5e-3, step)
        m64 = 0.9 * m64 + 0.1 * g
        v64 = 0.999 * v64 + 0.001 * g.astype(np.float64) ** 2
        alpha = 5e-3 * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        p64 = p64 - alpha * m64 / (np.sqrt(v64) + 1e-7)
    np.testing.assert_allclose(pd.cpu().numpy(), p64, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("synth", [None, dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16),
                                   dict(cls="TwoLayerSynthesis", channels=(24, 3), strides=(8, 2), kernel_sizes=(13, 5),
                                        activation_type="igdn")], ids=["two_layer_res", "jpeg_like", "two_layer"])
def test_sga_loss_and_gradients(synth, dev):
    """GPU loss == oracle training-mode loss with the same Gumbel noise; GPU gradients == central finite
    differences of the float64 oracle loss."""
    from shallow_ntc_amd.common import data_lib
    model, w, tc = make_model(dev, synth)
    lam = 0.02
    ref_model = model_np.Model(tc, rd_lambda=lam)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 60, 64, seed=9))        # pads to 64 x 64
    model.initialize_itinf(x)
    assert model.itinf and model.global_step == 0
    z0 = model.latent_rvs.uq[0].loc.cpu().numpy().astype(np.float64)
    y0 = model.latent_rvs.uq[1].loc.cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(3)
    gz, gy = gumbel(rng, z0.shape), gumbel(rng, y0.shape)
    tau = 0.5
    r = model._sga.loss_and_grads(t(x, dev), t(z0, dev), t(y0, dev), tau, lam, noise_z=t(gz, dev), noise_y=t(gy, dev))

    def oracle_loss(z, y):
        return ref_model.frame_loss(w, x, (z, y), sga=dict(tau=tau, gumbel_z=gz.astype(np.float64), gumbel_y=gy.astype(np.float64)))

    ref = oracle_loss(z0, y0)
    n, h, wd, _ = x.shape
    bpp = (r["bits_z"].cpu().numpy().mean() + r["bits_y"].cpu().numpy().mean()) / (h * wd)
    mse = (r["sse"].cpu().numpy() / (h * wd * 3)).mean()
    assert abs(bpp - ref["bpp"]) < 2e-5 * max(1, ref["bpp"])
    assert abs(mse - ref["mse"]) < 2e-5 * ref["mse"]
    g_z, g_y = r["g_z"].cpu().numpy(), r["g_y"].cpu().numpy()
    hstep = 1e-4
    checked = 0
    for arr, grad, which in ((z0, g_z, 0), (y0, g_y, 1)):
        flat_idx = rng.choice(arr.size, size=12, replace=False)
        for fi in flat_idx:
            idx = np.unravel_index(fi, arr.shape)
            if abs(arr[idx] - np.rint(arr[idx])) < 5e-3:
                continue
            ap, am = arr.copy(), arr.copy()
            ap[idx] += hstep
            am[idx] -= hstep
            lp = oracle_loss(ap, y0)["rd_loss"] if which == 0 else oracle_loss(z0, ap)["rd_loss"]
            lm = oracle_loss(am, y0)["rd_loss"] if which == 0 else oracle_loss(z0, am)["rd_loss"]
            fd = (lp - lm) / (2 * hstep)
            assert abs(grad[idx] - fd) <= 2e-3 * abs(fd) + 2e-6, (which, idx, grad[idx], fd)
            checked += 1
    assert checked >= 16


def test_sga_optimisation_improves_rd(dev):
    """itinf_on_data_batch in miniature (common/itinf_lib.py:26-93): the training objective decreases and the
    hard-rounded validation loss does not get worse than the starting point."""
    from shallow_ntc_amd.common import data_lib
    model, w, tc = make_model(dev, lam=0.05)
    x = data_lib.normalize_image(data_lib.synthetic_images(2, 64, 64, seed=4))
    before = model.validation_step(x).scalars_float
    model.initialize_itinf(x)
    first = None
    for step in range(150):
        m = model.itinf_train_step(x, seed=11).scalars_float
        first = first if first is not None else m
        assert np.isfinite(m["rd_loss"])
    assert model.global_step == 150 and abs(m["tau"] - 0.5) < 1e-12        # tau stays at tau_ub until t0 = 200
    assert m["rd_loss"] < first["rd_loss"]
    after = model.itinf_validation_step(x).scalars_float
    assert after["rd_loss"] < before["rd_loss"], (before, after)
    assert {"rd_loss", "bpp", "mse", "psnr", "tau", "scheduled_lr", "sched_rd_lambda"} <= set(m)
    assert abs(m["scheduled_lr"] - 5e-3) < 1e-12


@pytest.mark.parametrize("ch,n,hh,wh", [(12, 2, 9, 13), (24, 1, 16, 8), (48, 1, 5, 7)])
def test_out_layer_adjoint_stream_kernel(ch, n, hh, wh, dev):
    """sntc_two_layer_out_adjoint == the gather-GEMM adjoint plan (Conv2D 5x5/2 on the same kernel array) == float64 autograd."""
    from shallow_ntc_amd import ops
    from oracle import train_ref
    rng = np.random.default_rng(ch)
    w2 = (rng.standard_normal((5, 5, 3, ch)) * 0.2).astype(np.float32)          # Conv2DTranspose kernel [kh,kw,Cout=3,Cin=ch]
    g = rng.standard_normal((n, 2 * hh, 2 * wh, 3)).astype(np.float32)
    wd, gd = torch.from_numpy(w2).to(dev), torch.from_numpy(g).to(dev)
    got = ops.two_layer_out_adjoint(gd, wd, ch).cpu().numpy()
    plan = ops.ConvPlan("conv", wd, None, 2)
    via_gemm = plan(gd).cpu().numpy()
    h = torch.zeros((n, ch, hh, wh), dtype=torch.float64, requires_grad=True)
    out = train_ref.conv2d_transpose(h, torch.from_numpy(w2.astype(np.float64)), None, 2)
    (out * train_ref.as_input(g)).sum().backward()
    ref = h.grad.permute(0, 2, 3, 1).numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 2e-5 * scale and np.abs(via_gemm - ref).max() < 2e-5 * scale


@pytest.mark.parametrize("name", ["two_layer_syn2", "jpegl"])
def test_sga_gradients_at_full_width_against_float64_autograd(name, dev):
    """BASELINE configs[4] / [3] at their REAL widths (320-channel latents, 320 -> 320 -> 480 -> 640 hyper-synthesis, 13x13 / 8
    two-layer or 18x18 / 16 JPEG-like synthesis), one 128 x 128 image: the loss and EVERY component of d loss / d z_loc and
    d loss / d y_loc (1,280 + 20,480 values) against float64 autograd of the oracle loss under the same Gumbel noise -- not a
    few finite-difference samples on a reduced net."""
    from oracle import train_ref
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    cfg = configs.CONFIGS[name](rd_lambda=0.02)
    cfg.update(configs.itinf())
    model = Model(device=dev, quality_metrics=False, **cfg)
    w = dict(model.get_weights())
    rng = np.random.default_rng(8)
    for k in list(w):                                        # every path carries gradient: biases, spread scales, live prior
        if k.endswith("/bias"):
            w[k] = (0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif k.endswith("/beta"):
            w[k] = (1 + 0.5 * rng.random(w[k].shape)).astype(np.float32)
        elif k.startswith("prior/"):
            w[k] = (w[k] + 0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[320:] = rng.uniform(-1, 2.5, size=320)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    model.set_weights(w)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 128, 128, seed=3))
    z0 = (2.0 * rng.standard_normal((1, 2, 2, 320))).astype(np.float32)
    y0 = (3.0 * rng.standard_normal((1, 8, 8, 320))).astype(np.float32)
    gz, gy = gumbel(rng, z0.shape), gumbel(rng, y0.shape)
    tau, lam = 0.4, 0.02
    model.initialize_itinf(x)
    r = model._sga.loss_and_grads(t(x, dev), t(z0, dev), t(y0, dev), tau, lam, noise_z=t(gz, dev), noise_y=t(gy, dev))
    ref = train_ref.sga_loss_and_grads(cfg["transform_config"], w, x, z0, y0, tau, gz, gy, lam)
    bpp = (r["bits_z"].cpu().numpy().mean() + r["bits_y"].cpu().numpy().mean()) / (128 * 128)
    mse = (r["sse"].cpu().numpy() / (128 * 128 * 3)).mean()
    assert abs(bpp - ref["bpp"]) < 2e-5 * ref["bpp"] and abs(mse - ref["mse"]) < 2e-5 * ref["mse"], (bpp, ref["bpp"], mse, ref["mse"])
    for got, want, label in ((r["g_z"].cpu().numpy(), ref["g_z"], "z"), (r["g_y"].cpu().numpy(), ref["g_y"], "y")):
        assert np.abs(want).max() > 0
        # float32 kernels against float64 autograd: relative to the tensor's scale; the steep SGA derivative near integers
        # amplifies float32 rounding of (mu - floor mu), hence the per-element relative term
        err = np.abs(got - want)
        assert (err <= 2e-4 * np.abs(want) + 2e-5 * np.abs(want).max()).all(), (label, float(err.max()), float(np.abs(want).max()))

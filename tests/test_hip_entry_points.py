"""The reference-shaped entry points around the training=True branch (-m gpu), each against the oracle:
``UQLatentRV.sample / quantize`` (reference common/latent_rvs_lib.py:77-116),
``Model.frame_loss_given_latent_rvs(training=True)`` for the unoise / mixedq / sga methods (mshyper/models.py:234-359) and
for the factorized-prior model (factorized/models.py:89-183), and SGA iterative inference on the factorized-prior model
(factorized/models.py:108-118; mshyper/models.py:389-413 is shared)."""
import math

import numpy as np
import pytest
import torch

from oracle import model_np, train_ref
from oracle import ops_np as O

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def gumbel(rng, shape):
    return (-np.log(-np.log(rng.uniform(1e-6, 1 - 1e-6, size=shape + (2,))))).astype(np.float32)


def test_uq_latent_rv_sample_and_quantize(dev):
    from shallow_ntc_amd.common.latent_rvs_lib import LatentRVCollection, UQLatentRV
    rng = np.random.default_rng(0)
    c = 8
    loc = (3 * rng.standard_normal((2, 5, 7, c))).astype(np.float32)
    loc[0, 0, 0, :4] = [0.5, 1.5, -2.5, 3.0]                               # ties: round half to even
    hyper = rng.standard_normal((2, 5, 7, 2 * c)).astype(np.float32)
    mu = hyper[..., :c]
    per_ch = rng.standard_normal(c).astype(np.float32)
    rv = UQLatentRV(t(loc, dev))
    hd = t(hyper, dev)
    # training=False / quantize: round(loc - offset) + offset for every offset form the reference passes
    np.testing.assert_array_equal(rv.sample(False).cpu().numpy(), np.rint(loc))
    np.testing.assert_array_equal(rv.quantize().cpu().numpy(), np.rint(loc))
    ref = (np.rint(loc - mu) + mu).astype(np.float32)
    np.testing.assert_array_equal(rv.sample(False, "sga", offset=hd[..., :c], tau=0.3).cpu().numpy(), ref)      # a strided view
    np.testing.assert_array_equal(rv.quantize(t(mu, dev)).cpu().numpy(), ref)
    np.testing.assert_array_equal(rv.quantize(t(per_ch, dev)).cpu().numpy(), (np.rint(loc - per_ch) + per_ch).astype(np.float32))
    # unoise
    u = rng.uniform(-0.5, 0.5, size=loc.shape).astype(np.float32)
    np.testing.assert_array_equal(rv.sample(True, "unoise", noise=t(u, dev)).cpu().numpy(), loc + u)
    a = rv.sample(True, "unoise", seed=3, step=5).cpu().numpy() - loc
    b = rv.sample(True, "unoise", seed=3, step=5).cpu().numpy() - loc
    d = rv.sample(True, "unoise", seed=3, step=6).cpu().numpy() - loc
    assert np.array_equal(a, b) and not np.array_equal(a, d) and np.abs(a).max() < 0.5 + 1e-6 and abs(a.mean()) < 0.05
    # sga around an offset == the oracle's sga_round with the same Gumbel noise
    g = gumbel(rng, loc.shape)
    for tau in (0.5, 0.1):
        got = rv.sample(True, "sga", offset=hd[..., :c], noise=t(g, dev), tau=tau, tau_r=5e-4, tau_ub=0.5, tau_t0=200).cpu().numpy()
        want = O.sga_round(loc.astype(np.float64), tau, g.astype(np.float64), offset=mu.astype(np.float64))
        np.testing.assert_allclose(got, want, atol=3e-5)
    got = rv.sample(True, "sga", noise=t(g, dev), tau=0.4).cpu().numpy()
    np.testing.assert_allclose(got, O.sga_round(loc.astype(np.float64), 0.4, g.astype(np.float64)), atol=3e-5)
    # soft_round (tfc.soft_round): m + tanh(alpha r) / (2 tanh(alpha / 2)), identity for alpha < 1e-3
    for alpha in (0.5, 4.0, 20.0):
        x = loc.astype(np.float64) - mu
        m = np.floor(x) + 0.5
        want = m + np.tanh(alpha * (x - m)) / (2 * np.tanh(alpha / 2)) + mu
        got = rv.sample(True, "soft_round", offset=t(mu, dev), alpha=alpha).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=2e-5)
    np.testing.assert_array_equal(rv.sample(True, "soft_round", alpha=1e-4).cpu().numpy(), loc)
    with pytest.raises(NotImplementedError):
        rv.sample(True, "no_such_method")
    # the collection samples every rv with the kind's config (:137-155)
    col = LatentRVCollection(uq=(UQLatentRV(t(loc, dev)), UQLatentRV(t(2 * loc, dev))))
    s = col.sample(False, dict(uq=dict(method="unoise")))
    np.testing.assert_array_equal(s.uq[1].cpu().numpy(), np.rint(2 * loc))
    assert [tuple(v.shape) for v in col.trainable_variables] == [loc.shape, loc.shape]


SYN = dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn", res_type="conv")


def _mshyper_model(dev, uq, synthesis=SYN, analysis=None, extra=None):
    from shallow_ntc_amd.mshyper.models import Model
    cfg = dict(analysis=analysis or dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)), synthesis=synthesis)
    lc = dict(uq=dict(method=uq, **(extra or {})))
    model = Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=1000, latent_config=lc,
                  optimizer_config=dict(learning_rate=1e-3, global_clipnorm=None, warmup_steps=0), quality_metrics=False,
                  offset_heuristic=False)
    w = dict(model.get_weights())
    rng = np.random.default_rng(11)
    for k, v in w.items():
        if k.endswith("/bias"):
            w[k] = (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
    w["hyper_analysis/layer_2/bias"] = (1.5 * rng.standard_normal(w["hyper_analysis/layer_2/bias"].shape)).astype(np.float32)
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[32:] = rng.uniform(-1, 2.5, size=32)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    model.set_weights(w)
    return model, cfg, w


@pytest.mark.parametrize("uq", ["unoise", "mixedq"])
def test_training_frame_loss_noise_proxies(uq, dev):
    """frame_loss_given_latent_rvs(training=True) == the float64 oracle training loss under the same uniform noise, and ==
    the forward value Trainer.loss_and_grads differentiates (same (seed, step) -> the same draw)."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.train import GDN_BETA_MIN, Trainer, gdn_raw
    model, cfg, w = _mshyper_model(dev, uq)
    n, h, wd = 2, 64, 128
    x = data_lib.normalize_image(data_lib.synthetic_images(n, h, wd, seed=5))
    rng = np.random.default_rng(7)
    nz = rng.uniform(-0.5, 0.5, size=(n, h // 64, wd // 64, 32)).astype(np.float32)
    ny = rng.uniform(-0.5, 0.5, size=(n, h // 16, wd // 16, 32)).astype(np.float32)
    lat = model.infer_latent_rvs(x)
    loss, metrics = model.frame_loss_given_latent_rvs(x, lat, training=True, noise=(t(nz, dev), t(ny, dev)))
    s = metrics.scalars_float
    params = dict(w)
    params["synthesis/act/beta"] = (gdn_raw(params["synthesis/act/beta"], GDN_BETA_MIN), GDN_BETA_MIN)
    params["synthesis/act/gamma"] = (gdn_raw(params["synthesis/act/gamma"], 0.0), 0.0)
    ref = train_ref.loss_and_grads(cfg, params, x, nz, ny, 0.02, gdn_raw_names=("synthesis/act/beta", "synthesis/act/gamma"), uq=uq)
    assert abs(s["bpp"] - ref["bpp"]) < 2e-5 * max(1.0, ref["bpp"]), (s["bpp"], ref["bpp"])
    assert abs(s["mse"] - ref["mse"]) < 2e-5 * ref["mse"], (s["mse"], ref["mse"])
    assert abs(loss - ref["loss"]) < 2e-5 * ref["loss"]
    assert abs(s["psnr"] - (-10 * (math.log(ref["mse"]) - 2 * math.log(255.0)) / math.log(10))) < 1e-3 or n > 1
    assert {"rd_loss", "bpp", "mse", "psnr", "scheduled_lr", "sched_rd_lambda"} <= set(s) and "msssim" not in s
    # generator path: the trainer's forward pass draws the same noise from (seed, step)
    tr = Trainer(model, seed=model._seed)
    out = tr.loss_and_grads(t(x, dev), model._scheduled_rd_lambda)
    _, m2 = model.frame_loss_given_latent_rvs(x, lat, training=True)
    bpp_tr = float(out["bits_z"].cpu().numpy().mean() / (h * wd) + out["bits_y"].cpu().numpy().mean() / (h * wd))
    assert abs(m2.scalars_float["bpp"] - bpp_tr) < 1e-6 * max(1.0, bpp_tr)
    mse_tr = float((out["sse"].cpu().numpy() / (h * wd * 3)).mean())
    assert abs(m2.scalars_float["mse"] - mse_tr) < 1e-5 * mse_tr
    # and the eval branch is untouched
    _, ev = model.frame_loss_given_latent_rvs(x, lat, training=False)
    assert ev.scalars_float["bpp"] != s["bpp"]


def test_training_frame_loss_sga_and_soft_round(dev):
    """The explicit-sampling branch (:260-268,285-291): 'sga' equals the oracle under the same Gumbel noise and the loss
    itinf_train_step reports for its first step; 'soft_round' samples through UQLatentRV.sample and prices the samples
    with the noisy densities."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd import ops
    extra = dict(tau_r=5e-4, tau_ub=0.5, tau_t0=200)
    model, cfg, w = _mshyper_model(dev, "sga", extra=extra)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 60, 64, seed=9))           # pads to 64 x 64
    lat = model.infer_latent_rvs(x)
    z0, y0 = (rv.loc.cpu().numpy().astype(np.float64) for rv in lat.uq)
    rng = np.random.default_rng(3)
    gz, gy = gumbel(rng, z0.shape), gumbel(rng, y0.shape)
    loss, metrics = model.frame_loss_given_latent_rvs(x, lat, training=True, noise=(t(gz, dev), t(gy, dev)))
    ref = model_np.Model(cfg, rd_lambda=0.02).frame_loss(w, x, (z0, y0), sga=dict(tau=0.5, gumbel_z=gz.astype(np.float64),
                                                                                 gumbel_y=gy.astype(np.float64)))
    s = metrics.scalars_float
    assert abs(s["bpp"] - ref["bpp"]) < 2e-5 * max(1, ref["bpp"]) and abs(s["mse"] - ref["mse"]) < 2e-5 * ref["mse"]
    assert abs(loss - ref["rd_loss"]) < 2e-5 * ref["rd_loss"] and s["tau"] == 0.5
    # the same (seed, step) as itinf_train_step's first step -> the same loss value
    model.initialize_itinf(x)
    _, m_gen = model.frame_loss_given_latent_rvs(x, model.latent_rvs, training=True, seed=11)
    m_step = model.itinf_train_step(x, seed=11)
    assert abs(m_gen.scalars_float["rd_loss"] - m_step.scalars_float["rd_loss"]) < 1e-6 * m_step.scalars_float["rd_loss"]
    # soft_round: deterministic samples; rate of those samples under the noisy densities
    model2, cfg2, w2 = _mshyper_model(dev, "soft_round", extra=dict(alpha=4.0))
    lat2 = model2.infer_latent_rvs(x)
    loss2, m2 = model2.frame_loss_given_latent_rvs(x, lat2, training=True)
    z_t = lat2.uq[0].sample(True, "soft_round", alpha=4.0)
    hyper = model2._hyper_synthesis(z_t)
    y_t = lat2.uq[1].sample(True, "soft_round", offset=hyper[..., :32], alpha=4.0)
    bits_z, _ = ops.noisy_factorized(model2._get_prior(), z_t)
    bits_y, _, _ = ops.noisy_normal(y_t, hyper)
    bpp = float((bits_z.cpu().numpy().mean() + bits_y.cpu().numpy().mean()) / (60 * 64))
    assert abs(m2.scalars_float["bpp"] - bpp) < 1e-6 * max(1, bpp) and np.isfinite(loss2)
    ms, bs, fs = model_np._prior_lists(w2)
    zt64 = z_t.cpu().numpy().astype(np.float64)
    ref_bits_z = O.deep_factorized_logprob(zt64, ms, bs, fs).sum(axis=(1, 2, 3)) / -math.log(2)
    np.testing.assert_allclose(bits_z.cpu().numpy(), ref_bits_z, rtol=2e-5)


def _factorized_model(dev, uq, synthesis, analysis, extra=None):
    from shallow_ntc_amd.factorized.models import Model
    cfg = dict(analysis=analysis, synthesis=synthesis)
    model = Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=3000,
                  latent_config=dict(uq=dict(method=uq, **(extra or {}))), offset_heuristic=False, quality_metrics=False,
                  optimizer_config=dict(learning_rate=5e-3, reduce_lr_after=0.9, reduce_lr_factor=0.1, global_clipnorm=None, warmup_until=0.0))
    w = dict(model.get_weights())
    rng = np.random.default_rng(3)
    for k, v in w.items():
        if k.endswith("/bias"):
            w[k] = (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
        elif k.endswith("/beta"):
            w[k] = (1.0 + 0.5 * rng.random(v.shape)).astype(np.float32)
        elif k.endswith("/gamma"):
            w[k] = (0.1 * np.eye(v.shape[0]) + 0.02 * rng.random(v.shape)).astype(np.float32)
        elif k.endswith("/kernel") and k.startswith("analysis/layer_2"):
            w[k] = (6.0 * v).astype(np.float32)                               # latents a few units wide: rounding matters
        elif k.startswith("prior/"):
            w[k] = (v + 0.2 * rng.standard_normal(v.shape)).astype(np.float32)
    model.set_weights(w)
    return model, cfg, w


BLS = (dict(cls="BLS2017Synthesis", num_filters=32), dict(cls="BLS2017Analysis", num_filters=32))


def test_factorized_training_frame_loss(dev):
    from shallow_ntc_amd.common import data_lib
    model, cfg, w = _factorized_model(dev, "unoise", *BLS)
    n, h, wd = 2, 64, 96
    x = data_lib.normalize_image(data_lib.synthetic_images(n, h, wd, seed=5))
    rng = np.random.default_rng(7)
    ny = rng.uniform(-0.5, 0.5, size=(n, h // 16, wd // 16, 32)).astype(np.float32)
    lat = model.infer_latent_rvs(x)
    loss, metrics = model.frame_loss_given_latent_rvs(x, lat, training=True, noise=(None, t(ny, dev)))
    ref = train_ref.loss_and_grads(cfg, dict(w), x, None, ny, 0.02, factorized=True)
    s = metrics.scalars_float
    assert abs(s["bpp"] - ref["bpp"]) < 2e-5 * max(1, ref["bpp"]) and abs(s["mse"] - ref["mse"]) < 2e-5 * ref["mse"]
    assert abs(loss - ref["loss"]) < 2e-5 * ref["loss"]


@pytest.mark.parametrize("which", ["bls2017", "jpeg_like"])
def test_factorized_sga_loss_gradients_and_optimisation(which, dev):
    """SGA on the factorized-prior model (factorized/models.py:108-118): loss == the oracle's training-mode loss with the same
    Gumbel noise, d loss / d y_loc == central finite differences of the float64 oracle loss (through the SignalConv2D + IGDN
    synthesis of bls2017, and through a Conv2DTranspose synthesis), and a short optimisation lowers the objective."""
    from shallow_ntc_amd.common import data_lib
    extra = dict(tau_r=5e-4, tau_ub=0.5, tau_t0=200)
    if which == "bls2017":
        model, cfg, w = _factorized_model(dev, "sga", *BLS, extra=extra)
    else:
        model, cfg, w = _factorized_model(dev, "sga", dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16), BLS[1], extra=extra)
    lam = 0.02
    ref_model = model_np.Model(cfg, rd_lambda=lam, factorized=True)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 60, 64, seed=9))          # pads to 64 x 64
    model.initialize_itinf(x)
    assert model.itinf and len(model.latent_rvs.uq) == 1 and len(model.itinf_trainable_variables) == 1
    y0 = model.latent_rvs.uq[0].loc.cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(3)
    gy = gumbel(rng, y0.shape)
    tau = 0.5
    r = model._sga.loss_and_grads(t(x, dev), None, t(y0, dev), tau, lam, noise_y=t(gy, dev))

    def oracle_loss(y):
        return ref_model.frame_loss(w, x, (y,), sga=dict(tau=tau, gumbel_y=gy.astype(np.float64)))

    ref = oracle_loss(y0)
    n, h, wd, _ = x.shape
    bpp = r["bits_y"].cpu().numpy().mean() / (h * wd)
    mse = (r["sse"].cpu().numpy() / (h * wd * 3)).mean()
    assert abs(bpp - ref["bpp"]) < 2e-5 * max(1, ref["bpp"]) and abs(mse - ref["mse"]) < 2e-5 * ref["mse"]
    g_y = r["g_y"].cpu().numpy()
    hstep, checked = 1e-4, 0
    for fi in rng.choice(y0.size, size=16, replace=False):
        idx = np.unravel_index(fi, y0.shape)
        if abs(y0[idx] - np.rint(y0[idx])) < 5e-3:
            continue
        yp, ym = y0.copy(), y0.copy()
        yp[idx] += hstep
        ym[idx] -= hstep
        fd = (oracle_loss(yp)["rd_loss"] - oracle_loss(ym)["rd_loss"]) / (2 * hstep)
        assert abs(g_y[idx] - fd) <= 2e-3 * abs(fd) + 2e-6, (idx, g_y[idx], fd)
        checked += 1
    assert checked >= 10
    # frame_loss_given_latent_rvs(training=True) is the same value; then the reference's itinf loop in miniature
    loss_t, _ = model.frame_loss_given_latent_rvs(x, model.latent_rvs, training=True, noise=(None, t(gy, dev)))
    assert abs(loss_t - ref["rd_loss"]) < 2e-5 * ref["rd_loss"]
    before = model.itinf_validation_step(x).scalars_float
    first = None
    for _ in range(120):
        m = model.itinf_train_step(x, seed=5).scalars_float
        first = first or m
        assert np.isfinite(m["rd_loss"])
    assert model.global_step == 120 and m["rd_loss"] < first["rd_loss"] and "tau" in m
    after = model.itinf_validation_step(x).scalars_float
    assert after["rd_loss"] <= before["rd_loss"] + 1e-6, (before, after)


def test_sga_through_gdn_synthesis_of_the_hyperprior_model(dev):
    """mbt2018 transforms under SGA (SignalConv2D up layers + IGDN in the synthesis): gradients vs finite differences."""
    from shallow_ntc_amd.common import data_lib
    extra = dict(tau_r=5e-4, tau_ub=0.5, tau_t0=200)
    model, cfg, w = _mshyper_model(dev, "sga", synthesis=dict(cls="MBT2018Synthesis", channels_base=32),
                                   analysis=dict(cls="MBT2018Analysis", channels_base=32, output_channels=32), extra=extra)
    lam = 0.02
    ref_model = model_np.Model(cfg, rd_lambda=lam)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 64, 64, seed=2))
    model.initialize_itinf(x)
    z0, y0 = (rv.loc.cpu().numpy().astype(np.float64) for rv in model.latent_rvs.uq)
    rng = np.random.default_rng(4)
    gz, gy = gumbel(rng, z0.shape), gumbel(rng, y0.shape)
    r = model._sga.loss_and_grads(t(x, dev), t(z0, dev), t(y0, dev), 0.5, lam, noise_z=t(gz, dev), noise_y=t(gy, dev))

    def oracle_loss(z, y):
        return ref_model.frame_loss(w, x, (z, y), sga=dict(tau=0.5, gumbel_z=gz.astype(np.float64), gumbel_y=gy.astype(np.float64)))["rd_loss"]

    g_y = r["g_y"].cpu().numpy()
    checked = 0
    for fi in rng.choice(y0.size, size=12, replace=False):
        idx = np.unravel_index(fi, y0.shape)
        if abs(y0[idx] - np.rint(y0[idx])) < 5e-3:
            continue
        yp, ym = y0.copy(), y0.copy()
        yp[idx] += 1e-4
        ym[idx] -= 1e-4
        fd = (oracle_loss(z0, yp) - oracle_loss(z0, ym)) / 2e-4
        assert abs(g_y[idx] - fd) <= 2e-3 * abs(fd) + 2e-6, (idx, g_y[idx], fd)
        checked += 1
    assert checked >= 8


@pytest.mark.timeout(300)
def test_rccl_code_path_on_one_gpu(dev):
    """backend "nccl" IS RCCL on ROCm.  A one-rank process group on the one GPU a box offers runs every collective of the
    path through it: describe_world, the device-tensor all-gather of the metric rows, the max-reduce of the timing and the
    bucketed asynchronous gradient all-reduce (the branches that otherwise only ever ran under gloo on the CPU)."""
    import os
    import socket
    import torch.distributed as dist
    from shallow_ntc_amd import distributed as D
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        rank, local_rank, world = D.init(backend="nccl")
        assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
        info = D.describe_world(dev)
        assert info["backend"] == "nccl (RCCL on ROCm)" and info["world"] == 1 and info["rccl_version"]
        assert info["devices"][0]["arch"].startswith("gfx950")
        D.barrier()
        rows = np.array([[0.5, 30.0, 65.0], [0.25, 28.0, 103.0], [1.0, 35.0, 20.5]])
        table = D.gather_rows(rows, [2, 0, 1], 3, device=dev)                         # buffers live on the GPU under nccl
        np.testing.assert_array_equal(table, rows[[1, 2, 0]])
        assert D.max_over_ranks(3.25, device=dev) == 3.25
        out = D.run_units(5, lambda u: [u, 2.0 * u], device=dev)
        np.testing.assert_array_equal(out, np.array([[u, 2.0 * u] for u in range(5)]))
        flat = torch.arange(1000, dtype=torch.float32, device=dev)
        red = D.BucketReducer(flat, dict(a=(0, 400), b=(400, 1000)))
        assert red.active
        red.launch("a")
        red.launch("b")
        assert len(red._handles) == 2                                               # two asynchronous RCCL all-reduces in flight
        assert red.finish() == 1.0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(flat.cpu().numpy(), np.arange(1000, dtype=np.float32))
    finally:
        D.shutdown()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert not dist.is_initialized()

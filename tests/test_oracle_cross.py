"""Two independent restatements (float64 NumPy tap loops vs float32 torch library convolutions with
explicit pad / crop arithmetic) must agree, and the oracle must reproduce its committed fixtures."""
from pathlib import Path

import numpy as np
import pytest

from oracle import model_np
from oracle import ops_np as O
from oracle import torch_ref as R
from oracle import transforms_np as T

GOLD = Path(__file__).parent / "golden"


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("k,s,h,w", [(5, 2, 12, 10), (5, 2, 11, 13), (3, 1, 7, 9), (1, 1, 5, 6), (9, 4, 16, 20), (9, 4, 15, 18)])
def test_conv_down(k, s, h, w):
    rng = np.random.default_rng(k * 100 + h)
    x = rng.standard_normal((2, h, w, 6)).astype(np.float32)
    wk = rng.standard_normal((k, k, 6, 8)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    a = O.conv2d(x, wk, b, s)
    assert a.shape == (2, -(-h // s), -(-w // s), 8)
    assert rel(R.to_nhwc(R.conv2d(R.as_input(x), wk, b, s)), a) < 2e-6
    a = O.signal_conv_down(x, wk, b, s)
    assert rel(R.to_nhwc(R.signal_conv_down(R.as_input(x), wk, b, s)), a) < 2e-6


@pytest.mark.parametrize("k,s,h,w", [(5, 2, 6, 5), (3, 1, 7, 4), (13, 8, 3, 4), (18, 16, 2, 3), (16, 16, 2, 2), (6, 4, 3, 3), (9, 4, 3, 3)])
def test_conv_up(k, s, h, w):
    rng = np.random.default_rng(k * 100 + s)
    x = rng.standard_normal((2, h, w, 6)).astype(np.float32)
    wk = rng.standard_normal((k, k, 8, 6)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    a = O.conv2d_transpose(x, wk, b, s)
    assert a.shape == (2, h * s, w * s, 8)
    assert rel(R.to_nhwc(R.conv2d_transpose(R.as_input(x), wk, b, s)), a) < 2e-6
    # conv-transpose is the adjoint of the SAME conv with the same kernel array
    yy = rng.standard_normal(a.shape)
    assert abs((O.conv2d_transpose(x, wk, None, s) * yy).sum() - (x * O.conv2d(yy, wk, None, s)).sum()) < 1e-8 * np.abs(yy).sum()
    if k % 2:
        wk2 = rng.standard_normal((k, k, 6, 8)).astype(np.float32)
        a = O.signal_conv_up(x, wk2, b, s)
        assert rel(R.to_nhwc(R.signal_conv_up(R.as_input(x), wk2, b, s)), a) < 2e-6


def test_signal_conv_differs_from_keras_same_by_one_sample():
    """stride-2 'same_zeros' is the Keras SAME result shifted by one input sample."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 12, 12, 2))
    wk = rng.standard_normal((5, 5, 2, 3))
    a, b = O.signal_conv_down(x, wk, None, 2), O.conv2d(x, wk, None, 2)
    assert not np.allclose(a, b)
    np.testing.assert_allclose(O.conv2d(x, wk, None, 2, pad=((2, 2), (2, 2))), a, atol=1e-12)


TRANSFORMS = [
    ("ElicAnalysis", dict(channels=(16, 16, 32, 32)), (1, 64, 64, 3)),
    ("ElicSynthesis", dict(channels=(16, 16, 16, 3), cin=32), (1, 2, 2, 32)),
    ("TwoLayerResSynthesis", dict(cin=32), (1, 3, 4, 32)),
    ("TwoLayerSynthesis", dict(cin=32), (1, 2, 3, 32)),
    ("JPEGLikeSynthesis", dict(kernel_size=18, strides=16, cin=32), (1, 2, 3, 32)),
    ("HyperAnalysis", dict(bottleneck_size=32), (1, 8, 12, 32)),
    ("HyperSynthesis", dict(bottleneck_size=32), (1, 2, 3, 32)),
    ("BLS2017Analysis", dict(num_filters=16), (1, 32, 32, 3)),
    ("BLS2017Synthesis", dict(num_filters=16), (1, 2, 2, 16)),
    ("MBT2018Analysis", dict(channels_base=16, output_channels=24, gdn_alpha=2, gdn_epsilon=0.5), (1, 32, 32, 3)),
    ("MBT2018Synthesis", dict(channels_base=16, cin=24), (1, 2, 2, 24)),
    ("CNNAnalysis", dict(channels_base=16, output_channels=24, activation_type="gdn"), (1, 32, 32, 3)),
    ("JPEGLikeHyperSynthesis", dict(bottleneck_size=16), (1, 2, 2, 16)),
    ("TwoLayerResSynthesis", dict(cin=32, res_type="d2s"), (1, 3, 2, 32)),                  # transforms.py:339-348
    ("JPEGLikeSynthesis", dict(kernel_size=18, strides=16, use_offset=True, cin=32), (1, 2, 3, 32)),      # :291-293
    ("TwoLayerSynthesis", dict(cin=32, channels=(16, 3), activation_type="relu"), (1, 2, 2, 32)),
]


def test_depth_to_space_is_tensorflows_dcr_order():
    """tf.nn.depth_to_space(x, 2), NHWC: out[n, 2i + dy, 2j + dx, c] = x[n, i, j, (2 dy + dx) C + c]; both backends."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 2, 12)).astype(np.float32)
    y = O.depth_to_space(x, 2)
    assert y.shape == (2, 6, 4, 3)
    for dy in range(2):
        for dx in range(2):
            np.testing.assert_array_equal(y[:, dy::2, dx::2, :], x[..., (2 * dy + dx) * 3:(2 * dy + dx + 1) * 3])
    np.testing.assert_array_equal(R.to_nhwc(R.depth_to_space(R.as_input(x), 2)), y)


@pytest.mark.parametrize("cls,kw,shape", TRANSFORMS, ids=[t[0] + str(i) for i, t in enumerate(TRANSFORMS)])
def test_transforms_agree(cls, kw, shape):
    rng = np.random.default_rng(len(cls))
    t = T.build(cls, **kw)
    p = T.init_params(t.param_shapes(), rng)
    for k in p:
        if k.endswith("bias"):
            p[k] = (0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    x = rng.standard_normal(shape).astype(np.float32)
    assert rel(R.to_nhwc(t(p, x, be=R)), t(p, x)) < 5e-6


def test_oracle_reproduces_ops_fixture():
    g = np.load(GOLD / "ops.npz")
    fns = {"conv": O.conv2d, "convT": O.conv2d_transpose, "sigdown": O.signal_conv_down, "sigup": O.signal_conv_up}
    tags = sorted({k.split("/")[0] for k in g.files if k.endswith("/meta")})
    assert len(tags) == 9
    for tag in tags:
        k, s, cin, cout = g[f"{tag}/meta"]
        y = fns[tag.split("_")[0]](g[f"{tag}/x"], g[f"{tag}/w"], g[f"{tag}/b"], int(s))
        np.testing.assert_allclose(y, g[f"{tag}/y"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.gdn(g["gdn/x"], g["gdn/beta"], g["gdn/gamma"], True), g["gdn/igdn1"], atol=1e-13)
    np.testing.assert_allclose(O.gdn(g["gdn/x"], g["gdn/beta"], g["gdn/gamma"], False, 2, 0.5), g["gdn/classic"], atol=1e-13)
    c = g["normal/y"].shape[-1]
    mu, raw = g["normal/hyper"][..., :c], g["normal/hyper"][..., c:]
    y_hat, bits, sym = O.scale_indexed_normal(g["normal/y"], mu, np.exp(raw.astype(np.float64)))
    np.testing.assert_array_equal(sym.astype(np.int32), g["normal/symbols"])
    np.testing.assert_allclose(bits, g["normal/bits"], rtol=1e-12)
    np.testing.assert_array_equal(O.floats_to_pixels(g["pix/x_hat"], False), g["pix/pixels"])
    np.testing.assert_array_equal(O.pad_images(g["pix/x"], 8).astype(np.float32), g["pix/padded8"])


def test_oracle_reproduces_model_fixture():
    import json
    g = np.load(GOLD / "model_two_layer_small.npz")
    tc = json.loads(str(g["config"]))
    m = model_np.Model(tc, rd_lambda=float(g["rd_lambda"]))
    p = {k[2:]: g[k] for k in g.files if k.startswith("w/")}
    assert set(p) == set(m.param_shapes())
    r = m.end_to_end(p, g["x"])
    assert abs(r["bpp"] - float(g["bpp"])) < 1e-12 and abs(r["psnr"] - float(g["psnr"])) < 1e-10
    np.testing.assert_array_equal(r["recon_pixels"], g["recon_pixels"])
    assert abs(r["rd_loss"] - (r["bpp"] + 0.02 * r["mse"])) < 1e-12
    # evaluate(): per-image generator (mshyper/models.py:415-433)
    outs = list(m.evaluate(p, np.concatenate([g["x"], g["x"]])))
    assert len(outs) == 2 and abs(outs[1]["bpp"] - float(g["bpp"])) < 1e-12


@pytest.mark.parametrize("h,w", [(176, 200), (161, 185)])
def test_ms_ssim_restatements_agree(h, w):
    """float64 separable NumPy vs float32 torch (2-D softmax kernel, depthwise conv, replicate pad + avg_pool)."""
    rng = np.random.default_rng(h)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.clip(np.rint(np.stack([128 + 70 * np.sin(xx / 11 + c) * np.cos(yy / 6 - c) for c in range(3)], -1)[None]
                        + rng.normal(0, 6, size=(2, h, w, 3))), 0, 255)
    b = np.clip(np.rint(a + rng.normal(0, 9, size=a.shape)), 0, 255)
    ref = O.ms_ssim(a, b)
    assert ref.shape == (2,) and np.all(ref > 0.5) and np.all(ref < 1)
    np.testing.assert_allclose(R.ms_ssim_torch(a, b), ref, rtol=2e-5)
    np.testing.assert_allclose(O.ms_ssim(a, a), 1.0, atol=1e-12)
    v, db = O.image_quality(a[:, :100, :120], b[:, :100, :120])          # both sides < 160: single-scale SSIM
    np.testing.assert_allclose(v, O.ssim(a[:, :100, :120], b[:, :100, :120]))
    np.testing.assert_allclose(db, -10 * np.log10(1 - v))


def test_float64_library_backend_equals_the_tap_loop_oracle():
    """model_np's eval forward with its transforms on oracle/train_ref (float64 library convolutions: the full-size
    oracle of tests/test_hip_e2e_parity.py) equals the NumPy tap-loop oracle, and ``force_symbols`` at the oracle's own
    symbols reproduces its numbers."""
    from oracle import model_np, train_ref
    tc = dict(analysis=dict(cls="ElicAnalysis", channels=(16, 16, 16, 32)),
              synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                             activation_type="igdn", res_type="conv"))
    m = model_np.Model(tc, rd_lambda=0.02)
    p = m.init_params(3)
    rng = np.random.default_rng(1)
    p["analysis/conv3/kernel"] = (p["analysis/conv3/kernel"] * 30).astype(np.float32)
    x = rng.uniform(-0.5, 0.5, (1, 50, 100, 3))
    a, b = m.end_to_end(p, x), m.end_to_end(p, x, be=train_ref)
    assert np.abs(a["symbols_y"]).max() >= 2
    np.testing.assert_array_equal(a["symbols_y"], b["symbols_y"])
    assert abs(a["bpp"] - b["bpp"]) < 1e-10 and abs(a["psnr"] - b["psnr"]) < 1e-10
    assert np.abs(a["recon"] - b["recon"]).max() < 1e-11
    f = m.frame_loss(p, x, m.infer_latents(p, x), force_symbols=a["symbols_y"])
    assert abs(f["bpp"] - a["bpp"]) < 1e-12 and abs(f["psnr"] - a["psnr"]) < 1e-12
    assert f["tie_distance"].shape == a["symbols_y"].shape and (f["tie_distance"] <= 0.5).all()
    # moving one symbol by one step changes the rate: the forced path really evaluates the given integers
    s2 = a["symbols_y"].copy()
    s2[0, 0, 0, 0] += 1
    assert abs(m.frame_loss(p, x, m.infer_latents(p, x), force_symbols=s2)["bpp"] - a["bpp"]) > 1e-6


def test_sga_autograd_oracle_equals_the_numpy_oracle():
    """oracle/train_ref.sga_loss_and_grads (float64 autograd of the SGA objective, the reference of the full-width gradient
    test) gives the NumPy oracle's loss value, and its gradients equal central finite differences of that oracle loss."""
    from oracle import model_np, train_ref
    tc = dict(analysis=dict(cls="ElicAnalysis", channels=(16, 16, 16, 32)),
              synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                             activation_type="igdn", res_type="conv"))
    m = model_np.Model(tc, rd_lambda=0.02)
    w = m.init_params(3)
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.5, 0.5, (1, 64, 64, 3))
    z0, y0 = 2 * rng.standard_normal((1, 1, 1, 32)), 3 * rng.standard_normal((1, 4, 4, 32))
    g = lambda s: -np.log(-np.log(rng.uniform(1e-6, 1 - 1e-6, size=s + (2,))))
    gz, gy = g(z0.shape), g(y0.shape)
    sga = dict(tau=0.4, gumbel_z=gz, gumbel_y=gy)
    a = m.frame_loss(w, x, (z0, y0), sga=sga)
    b = train_ref.sga_loss_and_grads(tc, w, x, z0, y0, 0.4, gz, gy, 0.02)
    assert abs(a["rd_loss"] - b["loss"]) < 1e-9 * abs(a["rd_loss"]) and abs(a["bpp"] - b["bpp"]) < 1e-9 * a["bpp"]
    h = 1e-5
    for idx in [(0, 1, 2, 3), (0, 3, 0, 17), (0, 0, 3, 31)]:
        yp, ym = y0.copy(), y0.copy()
        yp[idx] += h
        ym[idx] -= h
        fd = (m.frame_loss(w, x, (z0, yp), sga=sga)["rd_loss"] - m.frame_loss(w, x, (z0, ym), sga=sga)["rd_loss"]) / (2 * h)
        assert abs(fd - b["g_y"][idx]) <= 1e-5 * abs(fd) + 1e-9, (idx, fd, b["g_y"][idx])
    zp, zm = z0.copy(), z0.copy()
    zp[0, 0, 0, 5] += h
    zm[0, 0, 0, 5] -= h
    fd = (m.frame_loss(w, x, (zp, y0), sga=sga)["rd_loss"] - m.frame_loss(w, x, (zm, y0), sga=sga)["rd_loss"]) / (2 * h)
    assert abs(fd - b["g_z"][0, 0, 0, 5]) <= 1e-5 * abs(fd) + 1e-9

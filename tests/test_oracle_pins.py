"""The oracle against everything the reference publishes for this path (CPU, no GPU needed).

The reference has no tests and its arithmetic cannot run here, so these are the only pins there are:
parameter counts (results/all_params.csv), FLOPs/pixel (results/all_fpp.csv), notebook shapes, the
zero-input => bias property of the JPEG-like synthesis (vis_syn_filters.ipynb), the scale table
(mshyper/models.py:28-34), and the metric identities on published per-image rows."""
import csv
import io
import json
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import model_np
from oracle import ops_np as O
from oracle import transforms_np as T

GOLD = Path(__file__).parent / "golden"
PUB = json.loads((GOLD / "published_rows.json").read_text())


def _table(text):
    rows = list(csv.reader(io.StringIO(text)))
    return {r[0]: dict(zip(rows[0][1:], r[1:])) for r in rows[1:]}


PARAMS = _table(PUB["params_csv"])
FPP = _table(PUB["fpp_csv"])
H, W = 512, 768


def build_sets():
    elic = T.build("ElicAnalysis", channels=(192, 192, 192, 320))
    return {
        "2-layer syn. (proposed)": dict(f=elic, g=T.build("TwoLayerResSynthesis", cin=320),
                                        f_h=T.build("HyperAnalysis", bottleneck_size=320),
                                        g_h=T.build("HyperSynthesis", bottleneck_size=320)),
        "JPEG-like syn. (proposed)": dict(g=T.build("JPEGLikeSynthesis", kernel_size=18, strides=16, cin=320)),
        "Ballé 2017 Factorized Prior": dict(f=T.build("CNNAnalysis", channels_base=192, output_channels=320),
                                            g=T.build("CNNSynthesis", channels_base=192, output_channels=3, cin=320)),
        "Minnen 2018 Hyperprior": dict(
            f=T.build("CNNAnalysis", channels_base=192, output_channels=320, activation_type="gdn"),
            g=T.build("CNNSynthesis", channels_base=192, output_channels=3, cin=320, activation_type="igdn"),
            f_h=T.build("HyperAnalysis", bottleneck_size=320), g_h=T.build("HyperSynthesis", bottleneck_size=320)),
        "He 2022 ELIC": dict(f=elic, g=T.build("ElicSynthesis", channels=(192, 192, 192, 3), cin=320)),
    }


@pytest.mark.parametrize("method", list(build_sets()))
def test_parameter_counts_match_published_table(method):
    for col, t in build_sets()[method].items():
        want = PARAMS[method][col]
        assert want != "", (method, col)
        assert t.num_params() == int(float(want)), (method, col)


def test_flops_per_pixel_match_published_table():
    """2*MAC + one add per bias element reproduces the pure-conv entries exactly; entries with GDN /
    residual adds / sigmoid gates agree to < 0.2 % (the TF profiler also counts those element-wise ops)."""
    px = H * W

    def bias_adds(t, h, w):
        tot, cin = 0, t.cin
        def walk(layer, cin, h, w):
            nonlocal tot
            if isinstance(layer, T.Seq):
                for l in layer.layers:
                    cin, h, w = walk(l, cin, h, w)
                return cin, h, w
            f, cout, ho, wo = layer.flops(cin, h, w)
            if isinstance(layer, T.Conv) and layer.bias:
                tot += cout * ho * wo
            return cout, ho, wo
        walk(t.graph, cin, h, w)
        return tot

    g_h = T.build("HyperSynthesis", bottleneck_size=320)
    f_h = T.build("HyperAnalysis", bottleneck_size=320)
    jp = T.build("JPEGLikeSynthesis", kernel_size=18, strides=16, cin=320)
    row = FPP["2-layer syn. (proposed)"]
    assert (g_h.flops(H // 64, W // 64) + bias_adds(g_h, H // 64, W // 64)) / px == float(row["g_h"]) == 30354.6875
    assert (f_h.flops(H // 16, W // 16) + bias_adds(f_h, H // 16, W // 16)) / px == float(row["f_h"]) == 13451.640625
    assert (jp.flops(H // 16, W // 16) + bias_adds(jp, H // 16, W // 16)) / px == float(FPP["JPEG-like syn. (proposed)"]["g"]) == 2433.0
    elic = T.build("ElicAnalysis", channels=(192, 192, 192, 320))
    assert abs(elic.flops(H, W) / px / float(row["f"]) - 1) < 2e-3
    two = T.build("TwoLayerResSynthesis", cin=320)
    assert abs(two.flops(H // 16, W // 16) / px / float(row["g"]) - 1) < 2e-3
    assert two.flops(H // 16, W // 16) / px < 50e3                       # README.md:17-19 "< 50K FLOPs/pixel"
    cnn = T.build("CNNAnalysis", channels_base=192, output_channels=320)
    assert abs(cnn.flops(H, W) / px / float(FPP["Ballé 2017 Factorized Prior"]["f"]) - 1) < 1e-3
    t24 = T.build("TwoLayerSynthesis", cin=320)                           # get_flops.ipynb cell 29: 11,349.0
    assert abs(t24.flops(H // 16, W // 16) / px / 11349.0 - 1) < 3e-3 and t24.num_params() == 1300347


def test_notebook_shapes_and_zero_input_is_bias():
    """get_flops.ipynb cell 26: y (1,32,48,320), z (1,8,12,320), x_hat (1,512,768,3);
    vis_syn_filters.ipynb cells 29-49: JPEG-like synthesis of zeros[1,1,1,320] is (1,16,16,3) == bias."""
    m = model_np.Model(dict(analysis=dict(cls="ElicAnalysis", channels=(192, 192, 192, 320)),
                            synthesis=dict(cls="TwoLayerResSynthesis")))
    assert m.bottleneck == 320 and m.prior_channels == 320 and m.downsample_factor == 64
    assert m._spatial(m.analysis, 512) == 32 and m._spatial(m.analysis, 768) == 48
    assert m._spatial(m.hyper_analysis, 32) == 8 and m._spatial(m.hyper_analysis, 48) == 12
    jp = T.build("JPEGLikeSynthesis", kernel_size=18, strides=16, cin=320)
    rng = np.random.default_rng(0)
    p = T.init_params(jp.param_shapes(), rng)
    p["conv/bias"] = np.array([-0.00739, -0.04296, -0.08146], np.float32)
    out = jp(p, np.zeros((1, 1, 1, 320)))
    assert out.shape == (1, 16, 16, 3)
    np.testing.assert_allclose(out, np.broadcast_to(p["conv/bias"].astype(np.float64), out.shape))
    two = T.build("TwoLayerResSynthesis", cin=8)
    assert two(T.init_params(two.param_shapes(), rng), np.zeros((1, 2, 3, 8))).shape == (1, 32, 48, 3)


def test_scale_table_constants():
    """mshyper/models.py:28-34 and SURVEY.md Appendix C."""
    assert O.NUM_SCALES == 64 and O.SCALE_MIN == 0.11 and O.SCALE_MAX == 256.0
    assert abs(O.SCALE_FACTOR - 0.12305479932808384) < 1e-15
    np.testing.assert_allclose(O.scale_fn([0, 1, 10, 31.5, 63]), [0.11, 0.124404103379, 0.376542, 5.306599664569, 256.0], rtol=2e-6)
    kat = {0.0: [7.908418055e-06, 18.47694976, 378.4317401, 22677.58857],
           10.0: [0.2937470448, 3.441034068, 35.88577211, 1941.606578],
           31.5: [3.735669186, 3.761209505, 3.965532119, 13.95192272],
           63.0: [9.325748982, 9.325759989, 9.325848044, 9.330151732]}
    for idx, want in kat.items():
        got = -O.noisy_normal_logprob(np.array([0.0, 1.0, -3.0, 20.0]), O.scale_fn(idx)) / math.log(2)
        np.testing.assert_allclose(got, want, rtol=1e-8)


def test_noisy_normal_is_a_probability_mass():
    for s in [0.11, 0.5, 3.0, 40.0]:
        v = np.arange(-2000, 2001, dtype=np.float64)
        assert abs(np.exp(O.noisy_normal_logprob(v, s)).sum() - 1) < 1e-9


def test_deep_factorized_is_a_probability_mass():
    rng = np.random.default_rng(1)
    p = model_np.init_deep_factorized(4, rng, (3, 3))
    for k in p:
        p[k] = p[k] + 0.2 * rng.standard_normal(p[k].shape).astype(np.float32)
    ms, bs, fs = model_np._prior_lists(p)
    v = np.arange(-400, 401, dtype=np.float64)[:, None] * np.ones((1, 4))
    mass = np.exp(O.deep_factorized_logprob(v, ms, bs, fs)).sum(0)
    np.testing.assert_allclose(mass, 1.0, atol=1e-6)


@pytest.mark.parametrize("method", ["2-layer_syn", "JPEG-like_syn"])
def test_metric_identities_on_published_rows(method):
    """results/readme.md: rd_loss = bpp + lambda * mse (0-255 scale); psnr = 10 log10(255^2 / mse)."""
    for r in PUB[method]:
        assert abs(r["rd_loss"] - (r["bpp"] + r["rd_lambda"] * r["mse"])) < 2e-5 * max(1, r["rd_loss"])
        _, psnr = O.mse_psnr(np.array([[math.sqrt(r["mse"])]]), np.array([[0.0]]))
        assert abs(psnr[0] - r["psnr"]) < 2e-4


def test_rounding_and_pixel_semantics():
    np.testing.assert_array_equal(O.round_half_even(np.array([0.5, 1.5, 2.5, -0.5, -1.5])), [0, 2, 2, -0, -2])
    x = np.array([[[[-0.6, -0.5, 0.5, 0.7, 100.5 / 255 - 0.5]]]], np.float32)
    np.testing.assert_array_equal(O.floats_to_pixels(x, False).ravel()[:4], [0, 0, 255, 255])
    img = np.arange(2 * 5 * 7 * 3, dtype=np.float64).reshape(2, 5, 7, 3)
    p = O.pad_images(img, 4)
    assert p.shape == (2, 8, 8, 3)
    np.testing.assert_array_equal(p[:, 5, :7], img[:, 3])   # reflect: row h+0 mirrors row h-2
    np.testing.assert_array_equal(p[:, :, 7], p[:, :, 5])
    np.testing.assert_array_equal(O.unpad_images(p, img.shape), img)
    assert O.pad_images(img[:, :4, :4], 4) is not None and O.pad_images(img[:, :4, :4], 4).shape == (2, 4, 4, 3)


def test_sga_schedule_and_rounding_limits():
    """latent_rvs_utils.py:90-103 / mshyper/configs/itinf.py:39-41: tau = 0.5 until t0 = 200, then decays."""
    assert O.sga_tau(0, 5e-4, 0.5) == 0.5 and O.sga_tau(200, 5e-4, 0.5) == 0.5
    assert abs(O.sga_tau(2200, 5e-4, 0.5) - 0.5 * math.exp(-1.0)) < 1e-12
    assert O.sga_tau(10**7, 5e-4, 0.5) == 1e-8
    mu = np.array([0.2, 1.7, -0.4, 3.0])
    g = np.zeros(mu.shape + (2,))
    out = O.sga_round(mu, 1e-3, g)                          # tau -> 0: deterministic rounding to nearest
    np.testing.assert_allclose(out, [0.0, 2.0, -0.0, 3.0], atol=1e-6)
    out = O.sga_round(mu, 0.5, g, offset=np.full(mu.shape, 0.25))
    assert np.all(out >= np.floor(mu - 0.25) + 0.25 - 1e-12) and np.all(out <= np.ceil(mu - 0.25) + 0.25 + 1e-12)


def test_bitstream_golden_and_tables():
    """tests/golden/bitstream.npz freezes the wire format: the pure-Python coder reproduces its words and decodes them,
    and the product's host-side table construction (numpy, no GPU needed) is the same table set as the oracle's."""
    from pathlib import Path
    from oracle import rans_np
    g = np.load(Path(__file__).parent / "golden" / "bitstream.npz")
    tabs = rans_np.normal_tables()
    assert [len(f) for _, f in tabs] == g["table_sizes"].tolist() and [lo for lo, _ in tabs] == g["table_min"].tolist()
    assert np.concatenate([np.asarray(f) for _, f in tabs]).tolist() == g["table_freqs"].tolist()
    vals, tids = g["values"], g["table_ids"]
    E = vals.shape[1]
    for segs in (1, 3):
        eseg = -(-(-(-E // segs)) // 64) * 64
        words, off = g[f"words_s{segs}"].tolist(), 0
        lanes = int(g[f"lanes_s{segs}"])
        for i, ln in enumerate(g[f"lens_s{segs}"].tolist()):
            b, s = divmod(i, segs)
            sl = slice(s * eseg, min(E, (s + 1) * eseg))
            assert rans_np.encode_stream(vals[b, sl], tids[b, sl], tabs, lanes) == words[off:off + ln]
            assert rans_np.decode_stream(words[off:off + ln], tids[b, sl], tabs, lanes) == vals[b, sl].tolist()
            off += ln
    import __graft_entry__ as graft
    graft.load_package()
    from shallow_ntc_amd import entropy_coding as ec
    prod = ec.normal_tables()
    assert [(lo, list(map(int, f))) for lo, f in prod] == [(lo, list(f)) for lo, f in tabs]


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: a second, independent derivation of the two riskiest [DEP] semantics.  Everything below is worked
# by hand from the written definitions with scalar arithmetic in this file -- no oracle helper, no SciPy
# -- and THEN compared with oracle/ops_np.py.  It pins the oracle to the survey's reading of the dependency, not to TensorFlow:
# parity stays "unpinned" (README / DESIGN 2).
# ---------------------------------------------------------------------------------------------------------------------
SIG_X6 = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
SIG_W5 = [1.0, 10.0, 100.0, 1000.0, 10000.0]
# tfc.SignalConv2D(corr=True, strides_down=2, "same_zeros"), k = 5: the kernel is CENTRED, y[i] = sum_j w[j] x[2i + j - 2]:
#   y[0] = w2 x0 + w3 x1 + w4 x2            =   100 +  2000 + 30000                = 32100
#   y[1] = w0 x0 + w1 x1 + w2 x2 + w3 x3 + w4 x4 = 1 + 20 + 300 + 4000 + 50000      = 54321
#   y[2] = w0 x2 + w1 x3 + w2 x4 + w3 x5    =     3 +    40 +   500 + 6000          =  6543
SIG_DOWN = [32100.0, 54321.0, 6543.0]
# Keras Conv2D(padding="SAME") on the same numbers pads (1, 2): y[i] = sum_j w[j] x[2i + j - 1] -- one sample to the left:
#   y[0] = w1 x0 + w2 x1 + w3 x2 + w4 x3 = 10 + 200 + 3000 + 40000 = 43210;  y[1] = 1*2+10*3+100*4+1000*5+10000*6 = 65432;
#   y[2] = w0 x3 + w1 x4 + w2 x5 = 4 + 50 + 600 = 654
KERAS_DOWN = [43210.0, 65432.0, 654.0]
# tfc.SignalConv2D(corr=False, strides_up=2), k = 5, x = [1, 2, 3]: out[2i + j - 2] += x[i] w[j], length 6, no flip:
#   i = 0: (0) 100, (1) 1000, (2) 10000;  i = 1: (0) 2, (1) 20, (2) 200, (3) 2000, (4) 20000;  i = 2: (2) 3, (3) 30, (4) 300, (5) 3000
SIG_UP = [102.0, 1020.0, 10203.0, 2030.0, 20300.0, 3000.0]
# Keras Conv2DTranspose(SAME) has pt = (5 - 2) // 2 = 1: out[2i + j - 1] += x[i] w[j]:
#   i = 0: (0) 10, (1) 100, (2) 1000, (3) 10000;  i = 1: (1) 2, (2) 20, (3) 200, (4) 2000, (5) 20000;  i = 2: (3) 3, (4) 30, (5) 300
KERAS_UP = [10.0, 102.0, 1020.0, 10203.0, 2030.0, 20300.0]


def _line(vals, axis):
    a = np.asarray(vals, np.float64)
    return a.reshape((1, -1, 1, 1) if axis == 0 else (1, 1, -1, 1))


def _kernel(vals, axis):
    a = np.asarray(vals, np.float64)
    return a.reshape((-1, 1, 1, 1) if axis == 0 else (1, -1, 1, 1))


def _up_line(y, axis):
    """The line of an up-sampled [1, n, 1, 1] / [1, 1, n, 1] tensor: the stride also doubles the width-1 axis, whose second
    sample receives nothing from a width-1 kernel."""
    y = np.asarray(y)[0, :, :, 0]
    line, rest = (y[:, 0], y[:, 1:]) if axis == 0 else (y[0, :], y[1:, :])
    assert not rest.any()
    return line.tolist()


@pytest.mark.parametrize("axis", [0, 1])
def test_signal_conv_same_zeros_origins_by_hand(axis):
    """SURVEY.md A.3 against A.1 / A.2: the SignalConv2D layers (MBT2018 / BLS2017, reference common/transforms.py:101-175) sit one
    sample away from the Keras layers of the same kernel size -- on both axes, down and up."""
    x, w = _line(SIG_X6, axis), _kernel(SIG_W5, axis)
    assert O.signal_conv_down(x, w, None, 2).ravel().tolist() == SIG_DOWN
    assert O.conv2d(x, w, None, 2).ravel().tolist() == KERAS_DOWN
    x3 = _line(SIG_X6[:3], axis)
    assert _up_line(O.signal_conv_up(x3, w, None, 2), axis) == SIG_UP          # kernel [kh, kw, Cin, Cout], true convolution
    assert _up_line(O.conv2d_transpose(x3, w, None, 2), axis) == KERAS_UP      # kernel [kh, kw, Cout, Cin]
    # k = 9, stride 4 (BLS2017's first layer, transforms.py:101-104): a single tap w[j] = 1 reads x[4i + j - 4]
    x16 = _line(np.arange(1.0, 17.0), axis)
    for j in (0, 4, 8):
        w9 = np.zeros(9); w9[j] = 1.0
        want = [float(4 * i + j - 4 + 1) if 0 <= 4 * i + j - 4 < 16 else 0.0 for i in range(4)]
        assert O.signal_conv_down(x16, _kernel(w9, axis), None, 4).ravel().tolist() == want
    # and its mirror (transforms.py:123-134): x[i] lands on out[4i + j - 4]
    for j in (0, 4, 8):
        w9 = np.zeros(9); w9[j] = 1.0
        want = [0.0] * 16
        for i, v in enumerate((1.0, 2.0, 3.0, 4.0)):
            if 0 <= 4 * i + j - 4 < 16:
                want[4 * i + j - 4] = v
        assert _up_line(O.signal_conv_up(_line([1.0, 2.0, 3.0, 4.0], axis), _kernel(w9, axis), None, 4), axis) == want


def test_signal_conv_channel_order_by_hand():
    """SignalConv2D kernels are [kh, kw, Cin, Cout] in BOTH directions (Keras Conv2DTranspose is [kh, kw, Cout, Cin]): a 1 x 1
    kernel w[0, 0, ci, co] maps x[ci] to y[co] -- y = [1*1 + 2*100, 1*10 + 2*1000] = [201, 2010]."""
    x = np.array([1.0, 2.0]).reshape(1, 1, 1, 2)
    w = np.array([[1.0, 10.0], [100.0, 1000.0]]).reshape(1, 1, 2, 2)
    assert O.signal_conv_down(x, w, None, 1).ravel().tolist() == [201.0, 2010.0]
    assert O.signal_conv_up(x, w, None, 1).ravel().tolist() == [201.0, 2010.0]
    assert O.conv2d_transpose(x, w, None, 1).ravel().tolist() == [1.0 * 1 + 2.0 * 10, 1.0 * 100 + 2.0 * 1000]


def _log_sf_mills(z, terms=6):
    """log(1 - Phi(z)) for z >> 1 from the Mills-ratio series: phi(z) / z * (1 - 1/z^2 + 3/z^4 - 15/z^6 + ...)."""
    s = t = 1.0
    for k in range(1, terms):
        t *= -(2 * k - 1) / (z * z)
        s += t
    return -z * z / 2 - math.log(z * math.sqrt(2 * math.pi)) + math.log(s)


def test_noisy_normal_tail_selection_by_hand():
    """log P(v) = log[Phi((v + .5) / s) - Phi((v - .5) / s)], taken from the survival-function pair right of the
    median and from the cdf pair left of it.  Far tails through the Mills series (no SciPy), the centre through erf; a float64
    evaluation that took the WRONG pair at v = +20, s = 0.11 would return Phi(186.4) - Phi(177.3) = 1 - 1 = 0."""
    s0 = 0.11
    # right tail, v = +20: P = sf(19.5 / s) - sf(20.5 / s); left tail, v = -3: P = Phi(-2.5 / s) - Phi(-3.5 / s) = sf(2.5/s) - sf(3.5/s)
    for v, lo, hi, want in ((20.0, 19.5 / s0, 20.5 / s0, 22677.58857), (-3.0, 2.5 / s0, 3.5 / s0, 378.4317401)):
        ls_lo, ls_hi = _log_sf_mills(lo), _log_sf_mills(hi)
        by_hand = -(ls_lo + math.log1p(-math.exp(ls_hi - ls_lo))) / math.log(2)
        assert abs(by_hand - want) < 1e-6 * want                                         # SURVEY.md Appendix C's figures
        got = -float(O.noisy_normal_logprob(np.array(v), s0)) / math.log(2)
        assert abs(got - by_hand) < 1e-9 * by_hand
        assert abs(-float(O.noisy_normal_logprob(np.array(-v), s0)) / math.log(2) - by_hand) < 1e-9 * by_hand    # symmetric
    # centre: P(0) = erf(0.5 / (s sqrt 2)) = 1 - erfc(.)
    by_hand = -math.log1p(-math.erfc(0.5 / (s0 * math.sqrt(2)))) / math.log(2)
    assert abs(by_hand - 7.908418055e-06) < 1e-14
    assert abs(-float(O.noisy_normal_logprob(np.array(0.0), s0)) / math.log(2) - by_hand) < 1e-15
    # moderate sigma, both sides: P(+-2) at s = 3 from erfc differences
    p2 = 0.5 * (math.erfc(1.5 / (3 * math.sqrt(2))) - math.erfc(2.5 / (3 * math.sqrt(2))))
    for v in (2.0, -2.0):
        assert abs(float(O.noisy_normal_logprob(np.array(v), 3.0)) - math.log(p2)) < 1e-12
    # the index quirk (mshyper/models.py:274-279): sigma = 0.11 * exp(0.12305479932808384 * clamp(exp(raw), 0, 63))
    raw = math.log(10.0)
    _, bits, _ = O.scale_indexed_normal(np.array([[[[1.0]]]]), np.array([[[[0.0]]]]), np.exp(np.array([[[[raw]]]])))
    sig = 0.11 * math.exp(0.12305479932808384 * 10.0)
    p1 = 0.5 * (math.erfc(0.5 / (sig * math.sqrt(2))) - math.erfc(1.5 / (sig * math.sqrt(2))))
    assert abs(float(bits[0]) + math.log2(p1)) < 1e-12 and abs(float(bits[0]) - 3.441034068) < 1e-8


SOFTPLUS_ONE = math.log(math.e - 1.0)      # softplus(SOFTPLUS_ONE) = 1


def _affine_prior(c, bias_last):
    """DeepFactorized (1 -> 3 -> 3 -> 1) with every softplus(matrix entry) = 1, zero factors and only the last bias non-zero:
    logits(x) = 3 * 3 * x + bias_last = 9 x + B -- a logistic distribution, closed form."""
    ms = [np.full((c, 3, 1), SOFTPLUS_ONE), np.full((c, 3, 3), SOFTPLUS_ONE), np.full((c, 1, 3), SOFTPLUS_ONE)]
    bs = [np.zeros((c, 3)), np.zeros((c, 3)), np.full((c, 1), bias_last)]
    fs = [np.zeros((c, 3)), np.zeros((c, 3))]
    return ms, bs, fs


def test_deep_factorized_closed_form_by_hand():
    """h <- softplus(M) h + b; h <- h + tanh(a) * tanh(h) between layers; cdf = sigmoid(logits); the same sf / cdf
    pair selection as A.6.  With the affine prior above P(v) = sigmoid(9 (v + .5)) - sigmoid(9 (v - .5)):
      v = 0:  tanh(2.25)                                 -> 0.03205510711151 bits
      v = +5: sigmoid(-40.5) - sigmoid(-49.5)            -> (40.5 - log1p(-e^-9) + log1p(e^-40.5)) / ln 2 = 58.42932720970 bits
              (the cdf pair would give sigmoid(49.5) - sigmoid(40.5) = 1 - 1 = 0 in float64: the selection matters)
      v = -5: the mirror image, through the cdf pair."""
    ms, bs, fs = _affine_prior(2, 0.0)
    v = np.array([[0.0, 0.0], [5.0, -5.0]])
    bits = -O.deep_factorized_logprob(v, ms, bs, fs) / math.log(2)
    want0 = -math.log2(math.tanh(2.25))
    want5 = (40.5 - math.log1p(-math.exp(-9.0)) + math.log1p(math.exp(-40.5))) / math.log(2)
    assert abs(want0 - 0.03205510711151081) < 1e-15 and abs(want5 - 58.42932720970238) < 1e-12
    np.testing.assert_allclose(bits, [[want0, want0], [want5, want5]], rtol=1e-12)
    # a shifted median (B = -18: the median sits at v = 2): P(2) = tanh(2.25) again, P(0) = sigmoid(-13.5) - sigmoid(-22.5)
    ms, bs, fs = _affine_prior(1, -18.0)
    bits = -O.deep_factorized_logprob(np.array([[2.0], [0.0]]), ms, bs, fs) / math.log(2)
    want_m = (13.5 - math.log1p(-math.exp(-9.0)) + math.log1p(math.exp(-13.5))) / math.log(2)
    # sigmoid(-13.5) - sigmoid(-22.5) = e^-13.5 / (1 + e^-13.5) - e^-22.5 / (1 + e^-22.5); to first order in e^-13.5 as above
    np.testing.assert_allclose(bits.ravel(), [want0, want_m], rtol=2e-6)
    exact = -math.log2(1 / (1 + math.exp(13.5)) - 1 / (1 + math.exp(22.5)))
    assert abs(float(bits[1, 0]) - exact) < 1e-9
    # the non-linearity sits AFTER matrix + bias, gated by tanh(factor): factor_0 = atanh(1/2), x = 0.5:
    #   h1 = 0.5 + 0.5 tanh(0.5) = 0.7310585786300049 (x 3);  h2 = 3 h1 (factor_1 = 0);  logits = 3 h2 = 6.579527207670044
    ms, bs, fs = _affine_prior(1, 0.0)
    fs[0] = np.full((1, 3), math.atanh(0.5))
    assert abs(float(O.deep_factorized_logits(np.array([0.5]), ms, bs, fs)[0]) - 6.579527207670044) < 1e-13
    # quantisation offset 0: z = 2.5 rounds half to even -> 2, z = 3.5 -> 4 (tf.round), bits summed over the coding rank
    ms, bs, fs = _affine_prior(1, 0.0)
    zq, b = O.batched_deep_factorized(np.array([2.5, 3.5, -0.5]).reshape(1, 1, 3, 1), ms, bs, fs)
    assert zq.ravel().tolist() == [2.0, 4.0, -0.0]

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

"""image -> (bpp, PSNR) of the HIP path against the float64 oracle, END TO END FROM PIXELS, at full width and at
Kodak size, with the BASELINE.json tolerance written into every assertion (-m gpu).

Everything upstream of round(y - mu) runs in float32 on the GPU and in float64 in the oracle, so a symbol whose
y - mu lies within ~1e-5 of a rounding boundary can come out one step away.  These tests measure how often that happens
on a 393,216-pixel image (491,520 symbols) and what it does to the rate and to the PSNR (reference
mshyper/models.py:300-317).  What is asserted, exactly:

  * single-image cases and every set in the reference's PUBLISHED operating range (0.115 ... 1.31 bpp): the RAW
    comparison -- |d bpp| <= 1e-4 and |d PSNR| <= 1e-3 dB on whatever the GPU produced, no condition on the flips;
  * the one set far above that range (~2.3 bpp): an image whose hyper-latent z the float64 oracle ITSELF puts within
    2e-5 of a rounding tie (at most two per image) is compared a second time with z pinned to the GPU's value
    (oracle/model_np.py `force_z`), and THAT comparison is held to the bars -- an escape, counted and reported: the raw
    numbers of those images (|d bpp| up to 2e-4) are in the report, and `raw_images_within_bars` says how many images
    of each set meet the bars without it.

The oracle's transforms run on oracle/train_ref (library convolutions in float64; it equals the NumPy tap-loop oracle
to 1e-12, tests/test_oracle_cross.py), the entropy models and pixel maths on ops_np.  The oracle is this repository's
restatement of the reference (parity unpinned: the reference has no tests, tensors or weights).

The measured counts are also written to gpurun_out/e2e_parity*.json (when that directory exists); tools/parity_tables.py
renders DESIGN.md's and README.md's parity tables from the committed copies under profiles/.
"""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import model_np, train_ref

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
REPORT = {}


OPERATING_POINTS = {
    # y_std: std of the latents; raw: range of the sigma half of the hyper-synthesis bias (sigma = 0.11 exp(0.123 exp(raw)));
    # z_std / mu_std: None leaves the random-init hyperprior as it is (z_hat == 0, mu ~ 0)
    "low_rate": dict(y_std=0.3, raw=(1.6, 2.5), z_std=None, mu_std=None),      # ~1.2 bpp: top of the published R-D range
    "high_rate": dict(y_std=0.5, raw=(1.8, 3.0), z_std=1.5, mu_std=0.2),       # ~3 bpp, live hyperprior (mu, sigma vary)
}


def _model(name, dev, y_std, raw, z_std, mu_std):
    """Full-width model with random-init weights rescaled so that the codec works where the reference's published
    curves live (0.1 ... 1.4 bpp; random-init nets by themselves give |y| << 1 with sigma_min everywhere, or, with
    y scaled up alone, hundreds of bpp -- neither says anything about the 1e-4 bpp bar): latents with std y_std,
    predicted scales around them, optionally a live hyperprior.  Gains are set from the GPU's own statistics of one
    probe image; everything is deterministic."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, **configs.CONFIGS[name](rd_lambda=0.02))
    model._step = 10 ** 9
    w = dict(model.get_weights())
    rng = np.random.default_rng(11)
    c = w["hyper_synthesis/layer_2/bias"].shape[0] // 2
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[c:] = rng.uniform(raw[0], raw[1], size=c)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    probe = data_lib.normalize_image(data_lib.synthetic_images(1, 256, 256, seed=99))

    def rescale(prefix, gain, rows=None):
        k = w[prefix + "/kernel"].copy()
        if rows is None:
            k *= np.float32(gain)
            w[prefix + "/bias"] = (w[prefix + "/bias"] * np.float32(gain)).astype(np.float32)
        else:
            k[:, :, rows, :] *= np.float32(gain)          # Keras transposed kernel [kh, kw, Cout, Cin]
        w[prefix + "/kernel"] = k.astype(np.float32)
        model.set_weights(w)

    model.set_weights(w)
    rescale("analysis/conv3", y_std / float(model.infer_latent_rvs(probe).uq[1].loc.std()))
    if z_std is not None:
        rescale("hyper_analysis/layer_2", z_std / float(model.infer_latent_rvs(probe).uq[0].loc.std()))
        lat = model.infer_latent_rvs(probe)
        mu = model._rate_and_reconstruction(lat)["hyper"][..., :c]
        rescale("hyper_synthesis/layer_2", mu_std / float(mu.std()), rows=slice(0, c))
    return model, w


@pytest.mark.parametrize("name", ["two_layer_syn", "jpegl"])
@pytest.mark.parametrize("hw", [(256, 256), (512, 768)], ids=["256x256", "512x768"])
@pytest.mark.parametrize("point", list(OPERATING_POINTS))
def test_image_to_bpp_psnr_at_full_width(name, hw, point, dev):
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    model, w = _model(name, dev, **OPERATING_POINTS[point])
    x = data_lib.normalize_image(data_lib.synthetic_images(1, hw[0], hw[1], seed=5 + hw[0]))
    # ---- GPU: the product path, image -> metrics
    lat = model.infer_latent_rvs(x)
    r = model._rate_and_reconstruction(lat, want_symbols=True)
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    # ---- float64 oracle, image -> metrics
    ref_model = model_np.Model(configs.CONFIGS[name]()["transform_config"], rd_lambda=0.02)
    ref = ref_model.end_to_end(w, x, be=train_ref)
    sym, rsym = r["symbols"].cpu().numpy(), ref["symbols_y"]
    flips = int((sym != rsym).sum())
    zflips = int((r["z_hat"].cpu().numpy() != ref["z_hat"]).sum())
    px = model.decode(*model.encode(torch.from_numpy(x).to(dev))[:2], hw).cpu().numpy()
    pix_diff = int((px != ref["recon_pixels"].astype(np.uint8)).sum())
    rep = dict(symbols=int(sym.size), symbol_flips=flips, z_flips=zflips, pixel_values=int(px.size),
               pixel_code_diffs=pix_diff, bpp_hip=m["bpp"], bpp_f64=float(ref["bpp"]), d_bpp=m["bpp"] - float(ref["bpp"]),
               psnr_hip=m["psnr"], psnr_f64=float(ref["psnr"]), d_psnr=m["psnr"] - float(ref["psnr"]),
               y_std=float(lat.uq[1].loc.std()), y_abs_max=float(lat.uq[1].loc.abs().max()),
               symbols_nonzero=float((sym != 0).mean()), z_hat_nonzero=float((ref["z_hat"] != 0).mean()))
    REPORT[f"{name}/{point}/{hw[0]}x{hw[1]}"] = rep
    print(json.dumps({f"{name}/{point}/{hw[0]}x{hw[1]}": rep}))
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "e2e_parity.json").write_text(json.dumps(REPORT, indent=1))
    assert 0.1 <= rep["bpp_f64"] <= 6.0, rep         # the operating range the tolerance is stated for
    assert abs(rep["d_bpp"]) <= 1e-4, rep            # BASELINE.json north_star tolerance, no escape
    assert abs(rep["d_psnr"]) <= 1e-3, rep
    # sanity bounds only -- the tolerance above is the bar.  A hyper-latent within ~1e-6 of a rounding boundary may come out one
    # step away in float32 (any summation order can do that: this case flipped one of 30,720 when the first layer's order
    # changed in round 3); it moves mu / sigma over its receptive field, hence the y flips it drags along.
    assert zflips <= 2, rep
    assert flips <= sym.size * (1e-4 + 1e-4 * zflips), rep


GDN_CONFIGS = [("bls2017", True, "analysis/layer_2", 0.8), ("mbt2018", False, "analysis/layer_3", 0.5),
               ("two_layer_syn2", False, "analysis/layer_3", 0.4)]


def _gdn_model(name, factorized, last, y_std, dev, precision="fp32"):
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.factorized.models import Model as FModel
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    cfg = configs.CONFIGS[name](rd_lambda=0.02)
    model = (FModel if factorized else Model)(device=dev, **cfg)
    model._step = 10 ** 9
    w = dict(model.get_weights())
    rng = np.random.default_rng(3)
    if not factorized:
        b = w["hyper_synthesis/layer_2/bias"].copy()
        c = b.shape[0] // 2
        b[c:] = rng.uniform(1.6, 2.5, size=c)
        w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
        model.set_weights(w)
    probe = data_lib.normalize_image(data_lib.synthetic_images(1, 256, 256, seed=99))
    gain = np.float32(y_std / float(model.infer_latent_rvs(probe).uq[-1].loc.std()))
    w[last + "/kernel"] = (w[last + "/kernel"] * gain).astype(np.float32)
    if last + "/bias" in w:
        w[last + "/bias"] = (w[last + "/bias"] * gain).astype(np.float32)
    model.set_weights(w)
    if precision != "fp32":
        model = (FModel if factorized else Model)(device=dev, precision=precision, **cfg)
        model._step = 10 ** 9
        model.set_weights(w)
    return model, w, cfg


@pytest.mark.parametrize("name,factorized,last,y_std", GDN_CONFIGS, ids=[c[0] for c in GDN_CONFIGS])
def test_gdn_signal_conv_configs_image_to_bpp_psnr_at_full_width(name, factorized, last, y_std, dev):
    """BASELINE.json configs[0] (factorized/configs/bls2017.py: 256 filters, 9x9 / 4 SignalConv2D, GDN), configs[1]
    (mshyper/configs/mbt2018.py: 192 / 320, 5x5 / 2 SignalConv2D, GDN / IGDN) and the model of configs[4]
    (mshyper/configs/two_layer_syn2.py: CNNAnalysis 256 -> 320 with leaky_relu, TwoLayerSynthesis) at their real widths, 256 x 256: image -> (bpp, PSNR) of the HIP path against the float64 oracle end to end from pixels, BASELINE tolerance
    asserted unconditionally.  Random-init weights with the last analysis layer rescaled (and, with a hyperprior, the
    predicted scales lifted) so that the codec works at a published rate instead of at sigma_min."""
    from shallow_ntc_amd.common import data_lib
    model, w, cfg = _gdn_model(name, factorized, last, y_std, dev)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, 256, 256, seed=6))
    lat = model.infer_latent_rvs(x)
    r = model._rate_and_reconstruction(lat, want_symbols=True)
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    ref_model = model_np.Model(cfg["transform_config"], rd_lambda=0.02, factorized=factorized)
    ref = ref_model.end_to_end(w, x, be=train_ref)
    if factorized:                       # no integer symbols on this path: y_hat = round(y - median) + median, compared as values
        sym = r["y_hat"].cpu().numpy()
        flips = int((np.abs(sym - ref["y_hat"]) > 0.25).sum())
    else:
        sym = r["symbols"].cpu().numpy()
        flips = int((sym != ref["symbols_y"]).sum())
    rep = dict(symbols=int(sym.size), symbol_flips=flips, bpp_hip=m["bpp"], bpp_f64=float(ref["bpp"]), d_bpp=m["bpp"] - float(ref["bpp"]),
               psnr_hip=m["psnr"], psnr_f64=float(ref["psnr"]), d_psnr=m["psnr"] - float(ref["psnr"]))
    REPORT[f"{name}/256x256"] = rep
    print(json.dumps({name: rep}))
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "e2e_parity.json").write_text(json.dumps(REPORT, indent=1))
    assert 0.05 <= rep["bpp_f64"] <= 8.0, rep
    assert abs(rep["d_bpp"]) <= 1e-4, rep            # BASELINE.json north_star tolerance
    assert abs(rep["d_psnr"]) <= 1e-3, rep
    assert flips <= sym.size * 1e-4, rep


# ---------------------------------------------------------------------------------------------------------------------------
# Where the reference's published curves live: 0.115 ... 1.31 bpp on Kodak (results/kodak/aggregate.json).  There sigma sits
# near SCALE_MIN, almost every symbol is 0 and the far-tail log_ndtr series carries the rate of the few that are not.
# ---------------------------------------------------------------------------------------------------------------------------
PUBLISHED_RANGE = [  # config, (H, W), target bpp
    ("two_layer_syn", (512, 768), 0.12), ("two_layer_syn", (512, 768), 0.25), ("two_layer_syn", (512, 768), 0.5),
    ("jpegl", (512, 768), 0.25),
    ("two_layer_syn2", (1200, 1200), 0.25),        # Tecnick size: reflect-pads to 1216 (common/image_utils.py:41-66), crops back
]


def _model_at_bpp(name, dev, x, target):
    """Random-init model whose last analysis layer is rescaled (bisection on the GPU's own rate for image ``x``) until the
    codec sits at ``target`` bpp, with the predicted scale indexes exp(raw) in 0.2 ... 1.6, i.e. sigma = 0.113 ... 0.135."""
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    model = Model(device=dev, quality_metrics=False, **configs.CONFIGS[name](rd_lambda=0.02))
    model._step = 10 ** 9
    w = dict(model.get_weights())
    rng = np.random.default_rng(17)
    c = w["hyper_synthesis/layer_2/bias"].shape[0] // 2
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[c:] = rng.uniform(-1.5, 0.5, size=c)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    # a trained hyperprior spends ~0.01 - 0.05 bpp on z; the framework-default deep-factorized density (init_scale 10) is wide
    # and would cost 0.42 bpp for all-zero hyper-latents by itself: narrow it 40x (softplus(matrix_0) scales the input axis)
    m0 = w["prior/matrix_0"].astype(np.float64)
    w["prior/matrix_0"] = np.log(np.expm1(40.0 * np.log1p(np.exp(m0)))).astype(np.float32)
    last = "analysis/conv3" if "analysis/conv3/kernel" in w else "analysis/layer_3"
    k0, b0 = w[last + "/kernel"].copy(), w[last + "/bias"].copy()

    def rate(gain):
        w[last + "/kernel"] = (k0 * np.float32(gain)).astype(np.float32)
        w[last + "/bias"] = (b0 * np.float32(gain)).astype(np.float32)
        model.set_weights(w)
        return model.validation_step(x).scalars_float["bpp"]

    lo, hi = 1e-2, 1e4                                   # the rate is monotone in the gain
    for _ in range(40):
        mid = float(np.sqrt(lo * hi))
        r = rate(mid)
        if abs(r - target) <= 0.04 * target:
            break
        lo, hi = (mid, hi) if r < target else (lo, mid)
    return model, w


PUBLISHED_RANGE_P = [c + ("fp32",) for c in PUBLISHED_RANGE] + [
    ("two_layer_syn", (512, 768), 0.12, "bf16x3"), ("two_layer_syn", (512, 768), 0.5, "bf16x3")]


@pytest.mark.parametrize("name,hw,target,precision", PUBLISHED_RANGE_P,
                         ids=[f"{n}-{h}x{w}-{t}bpp" + ("" if p == "fp32" else "-" + p) for n, (h, w), t, p in PUBLISHED_RANGE_P])
def test_published_operating_range(name, hw, target, precision, dev):
    """The same unconditional bars (|d bpp| <= 1e-4, |d PSNR| <= 1e-3 dB, image -> metrics from pixels, float32 GPU against
    the float64 oracle) at ~0.12 / 0.25 / 0.5 bpp at Kodak size and on the padded Tecnick path."""
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    x = data_lib.normalize_image(data_lib.synthetic_images(1, hw[0], hw[1], seed=31 + hw[0]))
    model, w = _model_at_bpp(name, dev, x, target)
    if precision != "fp32":              # the same weights in split precision: same bars; and its bitstream round trip
        from shallow_ntc_amd.mshyper.models import Model
        model = Model(device=dev, quality_metrics=False, precision=precision, **configs.CONFIGS[name](rd_lambda=0.02))
        model._step = 10 ** 9
        model.set_weights(w)
        xd = torch.from_numpy(x).to(dev)
        blob = model.compress(xd)
        z_hat, symbols = model.encode(xd)[:2]
        assert torch.equal(model.decompress(blob), model.decode(z_hat, symbols, hw))
    lat = model.infer_latent_rvs(x)
    r = model._rate_and_reconstruction(lat, want_symbols=True)
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    ref_model = model_np.Model(configs.CONFIGS[name]()["transform_config"], rd_lambda=0.02)
    ref = ref_model.end_to_end(w, x, be=train_ref)
    sym, rsym = r["symbols"].cpu().numpy(), ref["symbols_y"]
    flips = int((sym != rsym).sum())
    hyper = r["hyper"].cpu().numpy()
    c = sym.shape[-1]
    idx = np.exp(hyper[..., c:].astype(np.float64))
    rep = dict(target_bpp=target, symbols=int(sym.size), symbol_flips=flips, bpp_hip=m["bpp"], bpp_f64=float(ref["bpp"]),
               d_bpp=m["bpp"] - float(ref["bpp"]), psnr_hip=m["psnr"], psnr_f64=float(ref["psnr"]), d_psnr=m["psnr"] - float(ref["psnr"]),
               symbols_nonzero=float((sym != 0).mean()), symbols_abs_max=int(np.abs(sym).max()),
               scale_index_range=[float(idx.min()), float(idx.max())], padded=[int(v) for v in lat.uq[1].loc.shape[1:3]])
    key = f"{name}/{hw[0]}x{hw[1]}/{target}bpp" if precision == "fp32" else f"{precision}/{name}/{hw[0]}x{hw[1]}/{target}bpp"
    REPORT[key] = rep
    print(json.dumps({key: rep}))
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "e2e_parity.json").write_text(json.dumps(REPORT, indent=1))
    assert 0.9 * target <= rep["bpp_f64"] <= 1.1 * target, rep           # the case really sits in the published range
    assert rep["scale_index_range"][1] < 3.0 and rep["symbols_nonzero"] < 0.2, rep      # sigma near SCALE_MIN, mostly zeros
    assert abs(rep["d_bpp"]) <= 1e-4, rep                                # BASELINE.json north_star tolerance, no escape
    assert abs(rep["d_psnr"]) <= 1e-3, rep
    assert flips <= max(2, sym.size * 1e-5), rep
    if hw[0] % 64:                                                       # the Tecnick path: the latents live on the padded grid
        assert rep["padded"] == [-(-hw[0] // 64) * 4, -(-hw[1] // 64) * 4], rep


# ---------------------------------------------------------------------------------------------------------------------------
# Split precision (Model(precision="bf16x3")) through the same end-to-end check, all five BASELINE configs at
# their real widths: not bit-identical to fp32, held to the same bars against the float64 oracle.
# ---------------------------------------------------------------------------------------------------------------------------
BF3_CASES = [("two_layer_syn", (512, 768)), ("jpegl", (256, 256)), ("bls2017", (256, 256)), ("mbt2018", (256, 256)),
             ("two_layer_syn2", (256, 256))]


@pytest.mark.parametrize("name,hw", BF3_CASES, ids=[c[0] for c in BF3_CASES])
def test_bf16x3_image_to_bpp_psnr_at_full_width(name, hw, dev):
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model
    gdn = {c[0]: c for c in GDN_CONFIGS}
    factorized = False
    if name in gdn:
        _, factorized, last, y_std = gdn[name]
        model, w, cfg = _gdn_model(name, factorized, last, y_std, dev, precision="bf16x3")
    else:
        _, w = _model(name, dev, **OPERATING_POINTS["low_rate"])
        cfg = configs.CONFIGS[name](rd_lambda=0.02)
        model = Model(device=dev, precision="bf16x3", **cfg)
        model._step = 10 ** 9
        model.set_weights(w)
    x = data_lib.normalize_image(data_lib.synthetic_images(1, hw[0], hw[1], seed=5 + hw[0]))
    lat = model.infer_latent_rvs(x)
    r = model._rate_and_reconstruction(lat, want_symbols=True)
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    ref_model = model_np.Model(cfg["transform_config"], rd_lambda=0.02, factorized=factorized)
    ref = ref_model.end_to_end(w, x, be=train_ref)
    if factorized:
        sym = r["y_hat"].cpu().numpy()
        flips = int((np.abs(sym - ref["y_hat"]) > 0.25).sum())
    else:
        sym = r["symbols"].cpu().numpy()
        flips = int((sym != ref["symbols_y"]).sum())
    rep = dict(precision="bf16x3", symbols=int(sym.size), symbol_flips=flips, bpp_hip=m["bpp"], bpp_f64=float(ref["bpp"]),
               d_bpp=m["bpp"] - float(ref["bpp"]), psnr_hip=m["psnr"], psnr_f64=float(ref["psnr"]), d_psnr=m["psnr"] - float(ref["psnr"]))
    if not factorized:                   # the bitstream round trip in this arithmetic: the decoder rebuilds mu / sigma bit for bit
        xd = torch.from_numpy(x).to(dev)
        z_hat, symbols = model.encode(xd)[:2]
        px = model.decode(z_hat, symbols, hw).cpu().numpy()
        rep["pixel_code_diffs"] = int((px != ref["recon_pixels"].astype(np.uint8)).sum())
        rep["pixel_values"] = int(px.size)
    REPORT[f"bf16x3/{name}/{hw[0]}x{hw[1]}"] = rep
    print(json.dumps({f"bf16x3/{name}": rep}))
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "e2e_parity.json").write_text(json.dumps(REPORT, indent=1))
    assert 0.05 <= rep["bpp_f64"] <= 8.0, rep
    assert abs(rep["d_bpp"]) <= 1e-4, rep            # BASELINE.json north_star tolerance
    assert abs(rep["d_psnr"]) <= 1e-3, rep
    assert flips <= sym.size * 2e-4, rep


# ---------------------------------------------------------------------------------------------------------------------------
# The margin, not one image: all 24 Kodak-shaped images (18 landscape + 6 portrait, Kodak's own order) through the same
# end-to-end comparison at two operating points; every image is held to the bars by itself and the distribution
# (max / median |d bpp|, |d PSNR|, symbol flips, hyper-latent flips) is written out for DESIGN.md / README.md
# (tools/parity_tables.py renders those tables from the committed JSON -- no hand-typed parity number).
# ---------------------------------------------------------------------------------------------------------------------------
KODAK_SHAPES = [(768, 512) if i in (3, 8, 9, 16, 17, 18) else (512, 768) for i in range(24)]
SETS = {
    "2.3bpp": dict(config="two_layer_syn", point="high_rate", precision="fp32", shapes=KODAK_SHAPES),
    "0.25bpp": dict(config="two_layer_syn", point=0.25, precision="fp32", shapes=KODAK_SHAPES),
    "0.12bpp": dict(config="two_layer_syn", point=0.12, precision="fp32", shapes=KODAK_SHAPES[::2]),      # the published range's two ends
    "0.5bpp": dict(config="two_layer_syn", point=0.5, precision="fp32", shapes=KODAK_SHAPES[::2]),        # on every second image (12)
    "jpegl/0.25bpp": dict(config="jpegl", point=0.25, precision="fp32", shapes=KODAK_SHAPES),
    "bf16x3/0.25bpp": dict(config="two_layer_syn", point=0.25, precision="bf16x3", shapes=KODAK_SHAPES),
    "two_layer_syn2/5x1200x1200/0.25bpp": dict(config="two_layer_syn2", point=0.25, precision="fp32", shapes=[(1200, 1200)] * 5),
}
_ORACLE_CACHE = {}      # (config, point) -> per-image float64 results: the bf16x3 set runs the fp32 set's weights and images


@pytest.mark.parametrize("label", list(SETS))
def test_kodak_set_margin_distribution(label, dev):
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.mshyper import configs
    spec = SETS[label]
    name, point, shapes = spec["config"], spec["point"], spec["shapes"]
    images = [data_lib.normalize_image(data_lib.synthetic_images(1, h, w, seed=700 + i)) for i, (h, w) in enumerate(shapes)]
    if isinstance(point, str):
        model, w = _model(name, dev, **OPERATING_POINTS[point])
    else:
        model, w = _model_at_bpp(name, dev, images[0], point)
    if spec["precision"] != "fp32":          # the same weights in split precision: the same bars
        from shallow_ntc_amd.mshyper.models import Model
        model = Model(device=dev, quality_metrics=False, precision=spec["precision"], **configs.CONFIGS[name](rd_lambda=0.02))
        model._step = 10 ** 9
        model.set_weights(w)
    ref_model = model_np.Model(configs.CONFIGS[name]()["transform_config"], rd_lambda=0.02)
    # the float64 oracle on each image size's images as ONE batch (its library convolutions thread better that way);
    # every image is then read out by itself
    key = (name, point)
    if key not in _ORACLE_CACHE:
        oracle = {}
        for shape in sorted(set(shapes)):
            ids = [i for i, sh in enumerate(shapes) if sh == shape]
            xb = np.concatenate([images[i] for i in ids])
            rl = ref_model.infer_latents(w, xb, be=train_ref)
            rf = ref_model.frame_loss(w, xb, rl, be=train_ref)
            npix = float(shape[0] * shape[1])
            for k, i in enumerate(ids):
                oracle[i] = dict(lat=tuple(t[k:k + 1] for t in rl), symbols_y=rf["symbols_y"][k:k + 1], z_hat=rf["z_hat"][k:k + 1],
                                 bpp=float((rf["bits_z"][k] + rf["bits_y"][k]) / npix), psnr=float(rf["psnrs"][k]))
        _ORACLE_CACHE[key] = oracle
    oracle = _ORACLE_CACHE[key]
    pinned_allowed = isinstance(point, str)        # only the set far above the published range may pin a tied hyper-latent
    rows = []
    for i, x in enumerate(images):
        lat = model.infer_latent_rvs(x)
        r = model._rate_and_reconstruction(lat, want_symbols=True)
        _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
        m = metrics.scalars_float
        ref, rlat = oracle[i], oracle[i]["lat"]
        sym, z_gpu = r["symbols"].cpu().numpy(), r["z_hat"].cpu().numpy()
        zdiff = z_gpu != ref["z_hat"]
        row = dict(image=i, shape=list(shapes[i]), symbols=int(sym.size), symbol_flips=int((sym != ref["symbols_y"]).sum()),
                   z_flips=int(zdiff.sum()), bpp_f64=float(ref["bpp"]), d_bpp=m["bpp"] - float(ref["bpp"]),
                   psnr_f64=float(ref["psnr"]), d_psnr=m["psnr"] - float(ref["psnr"]))
        row["raw_within_bars"] = bool(abs(row["d_bpp"]) <= 1e-4 and abs(row["d_psnr"]) <= 1e-3 and row["symbol_flips"] <= sym.size * 1e-4)
        if row["z_flips"] and pinned_allowed:
            # a hyper-latent the ORACLE itself puts on a rounding tie (any float32 implementation, the reference's included, can
            # land on either side) moves mu / sigma over its receptive field: it must really be a tie, there may be at most two,
            # and with the hyper-latents pinned to the GPU's the image is held to the bars like every other
            pinned = ref_model.frame_loss(w, x, rlat, be=train_ref, force_z=z_gpu)
            row.update(z_tie_distance_max=float(pinned["z_tie_distance"][zdiff].max()),
                       symbol_flips_at_gpu_z=int((sym != pinned["symbols_y"]).sum()),
                       d_bpp_at_gpu_z=m["bpp"] - float(pinned["bpp"]), d_psnr_at_gpu_z=m["psnr"] - float(pinned["psnr"]))
            assert row["z_flips"] <= 2 and row["z_tie_distance_max"] <= 2e-5, row
            held = (row["d_bpp_at_gpu_z"], row["d_psnr_at_gpu_z"], row["symbol_flips_at_gpu_z"])
        else:
            held = (row["d_bpp"], row["d_psnr"], row["symbol_flips"])
        rows.append(row)
        # the BASELINE.json bars, per image (raw everywhere but on the tied images of the ~2.3 bpp set: module docstring)
        assert abs(held[0]) <= 1e-4, row
        assert abs(held[1]) <= 1e-3, row
        assert held[2] <= sym.size * 1e-4, row
    ab = lambda k: np.abs(np.array([r[k] for r in rows], np.float64))
    tied = [r for r in rows if "d_bpp_at_gpu_z" in r]          # compared a second time with z pinned (the ~2.3 bpp set only)
    clean = [r for r in rows if "d_bpp_at_gpu_z" not in r]     # held to the bars raw
    abc = lambda k, rs: np.abs(np.array([r[k] for r in rs], np.float64)) if rs else np.zeros(1)
    summary = dict(images=len(rows), config=name, precision=spec["precision"], image_sizes=sorted({f"{h}x{w}" for h, w in shapes}),
                   symbols_per_image=int(rows[0]["symbols"]), raw_images_within_bars=sum(r["raw_within_bars"] for r in rows),
                   bpp_f64_range=[min(r["bpp_f64"] for r in rows), max(r["bpp_f64"] for r in rows)],
                   images_without_z_flips=len(clean),
                   max_abs_d_bpp=float(abc("d_bpp", clean).max()), median_abs_d_bpp=float(np.median(abc("d_bpp", clean))),
                   max_abs_d_psnr=float(abc("d_psnr", clean).max()), median_abs_d_psnr=float(np.median(abc("d_psnr", clean))),
                   max_symbol_flips=int(abc("symbol_flips", clean).max()), total_symbol_flips=int(abc("symbol_flips", clean).sum()),
                   images_with_symbol_flips=int((abc("symbol_flips", clean) > 0).sum()),
                   share_of_the_bpp_bar_used=float(abc("d_bpp", clean).max() / 1e-4),
                   share_of_the_psnr_bar_used=float(abc("d_psnr", clean).max() / 1e-3),
                   images_with_z_flips=len(tied), max_z_flips=int(ab("z_flips").max()),
                   z_tie_distance_max=float(max((r["z_tie_distance_max"] for r in tied), default=0.0)),
                   raw_max_abs_d_bpp_with_z_flips=float(abc("d_bpp", tied).max()),
                   raw_max_symbol_flips_with_z_flips=int(abc("symbol_flips", tied).max()),
                   max_abs_d_bpp_at_gpu_z=float(abc("d_bpp_at_gpu_z", tied).max()),
                   max_abs_d_psnr_at_gpu_z=float(abc("d_psnr_at_gpu_z", tied).max()),
                   max_symbol_flips_at_gpu_z=int(abc("symbol_flips_at_gpu_z", tied).max()))
    out = ROOT / "gpurun_out"
    if out.is_dir():
        f = out / "e2e_parity_kodak24.json"
        rep = json.loads(f.read_text()) if f.exists() else {}
        rep[label] = dict(summary=summary, per_image=rows)
        f.write_text(json.dumps(rep, indent=1))
    print(json.dumps({label: summary}))
    if not pinned_allowed:                            # the set really sits around the published operating point
        assert 0.3 * point <= summary["bpp_f64_range"][0] and summary["bpp_f64_range"][1] <= 3.0 * point, summary

"""HIP kernels against the committed golden fixtures (tests/golden/*.npz) -- no oracle code needed
to produce the expected values on the GPU box (-m gpu)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def test_ops_fixture(dev):
    from shallow_ntc_amd import ops
    from shallow_ntc_amd.common._graph import GDN
    g = np.load(GOLD / "ops.npz")
    tags = sorted({k.split("/")[0] for k in g.files if k.endswith("/meta")})
    for tag in tags:
        k, s, cin, cout = g[f"{tag}/meta"]
        plan = ops.ConvPlan(tag.split("_")[0], t(g[f"{tag}/w"], dev), t(g[f"{tag}/b"], dev), int(s))
        assert rel(plan(t(g[f"{tag}/x"], dev)).cpu().numpy(), g[f"{tag}/y"]) < 2e-5, tag
    for name, inv, a, e in [("igdn1", True, 1, 1.0), ("gdn1", False, 1, 1.0), ("classic", False, 2, 0.5)]:
        node = GDN("g", inv, a, e)
        node.build({"g/beta": t(g["gdn/beta"], dev), "g/gamma": t(g["gdn/gamma"], dev)}, 12)
        assert rel(node(t(g["gdn/x"], dev)).cpu().numpy(), g[f"gdn/{name}"]) < 2e-6
    y_hat, bits, sym = ops.entropy_scale_normal(t(g["normal/y"], dev), t(g["normal/hyper"], dev), want_symbols=True)
    np.testing.assert_array_equal(sym.cpu().numpy(), g["normal/symbols"])          # integer domain: bit exact
    assert rel(bits.cpu().numpy(), g["normal/bits"]) < 2e-5
    assert rel(y_hat.cpu().numpy(), g["normal/y_hat"]) < 1e-6
    for tag, nl in [("df33", 3), ("df333", 4)]:
        prior = ops.DeepFactorizedPrior([g[f"{tag}/prior/matrix_{k}"] for k in range(nl)],
                                        [g[f"{tag}/prior/bias_{k}"] for k in range(nl)],
                                        [g[f"{tag}/prior/factor_{k}"] for k in range(nl - 1)])
        z_hat, zb = prior(t(g[f"{tag}/z"], dev))
        np.testing.assert_array_equal(z_hat.cpu().numpy(), g[f"{tag}/z_hat"])
        assert rel(zb.cpu().numpy(), g[f"{tag}/bits"]) < 2e-5
    sse, px = ops.pixels_sse(t(g["pix/x"], dev), t(g["pix/x_hat"], dev), want_pixels=True)
    np.testing.assert_array_equal(px.cpu().numpy(), g["pix/pixels"])               # uint8: bit exact
    assert abs(sse.cpu().numpy()[0] / g["pix/x"].size - g["pix/mse"][0]) < 1e-9
    np.testing.assert_array_equal(ops.pad_reflect(t(g["pix/x"], dev), 16, 8).cpu().numpy(), g["pix/padded8"])


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_model_fixture(precision, dev):
    """Reduced-width two_layer_syn (16-channel ELIC blocks also exercise the scalar-gather path).  Also in the split-precision
    mode: at this image size only the layers with at least one 256-row strip per image take it (common/_graph.py::DualPlan),
    the bars are the same; the full-width counterpart is tests/test_hip_e2e_parity.py::test_bf16x3_image_to_bpp_psnr_at_full_width."""
    from shallow_ntc_amd.mshyper.models import Model
    g = np.load(GOLD / "model_two_layer_small.npz")
    tc = json.loads(str(g["config"]))
    model = Model(rd_lambda=float(g["rd_lambda"]), transform_config=tc, device=dev, precision=precision)
    model._step = 10**9                      # past the lambda warm-up
    w = {k[2:]: g[k] for k in g.files if k.startswith("w/")}
    assert set(w) == set(model.get_weights())
    model.set_weights(w)
    x = g["x"]
    lat = model.infer_latent_rvs(x)
    assert rel(lat.uq[1].loc.cpu().numpy(), g["y"]) < 5e-5 and rel(lat.uq[0].loc.cpu().numpy(), g["z"]) < 5e-5
    r = model._rate_and_reconstruction(lat, want_symbols=True)
    sym = r["symbols"].cpu().numpy()
    flip = sym != g["symbols_y"]
    np.testing.assert_array_equal(r["z_hat"].cpu().numpy(), g["z_hat"])
    assert int(flip.sum()) <= 1
    _, metrics = model.frame_loss_given_latent_rvs(x, lat, training=False)
    m = metrics.scalars_float
    if not flip.any():                       # the frozen numbers themselves
        want_bpp, want_psnr = float(g["bpp"]), float(g["psnr"])
    else:                                    # a symbol within 1e-4 of a rounding boundary went the other way (3500-pixel
        # image: one symbol is ~5e-4 bpp): the oracle's rate / distortion AT THE GPU'S INTEGERS, from the frozen latents
        from oracle import model_np
        ref = model_np.Model(tc, rd_lambda=float(g["rd_lambda"])).frame_loss(w, x, (g["z"], g["y"]), force_symbols=sym)
        assert (ref["tie_distance"][flip] < 1e-4).all()
        want_bpp, want_psnr = ref["bpp"], ref["psnr"]
    assert abs(m["bpp"] - want_bpp) <= 1e-4            # BASELINE.json tolerances, both branches
    assert abs(m["psnr"] - want_psnr) <= 1e-3
    assert abs(m["rd_loss"] - (m["bpp"] + 0.02 * m["mse"])) < 1e-5

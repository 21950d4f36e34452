"""Framework-default initial weights put the codec at sigma_min with |y| << 1: every symbol is zero and nothing that acts on
the latents (SGA, the entropy coder) shows.  ``apply(model)`` rescales them -- deterministically, from the GPU's own
statistics of one probe image -- so that an UNTRAINED model operates near a published rate: latents with the given
standard deviation, predicted scales around them.  A plumbing aid for the drivers in tools/ when no checkpoint exists;
tests/test_hip_e2e_parity.py does the same for the parity cases."""
import numpy as np


def apply(model, y_std=0.4, raw=(1.6, 2.5), seed=11):
    from shallow_ntc_amd.common import data_lib
    w = dict(model.get_weights())
    rng = np.random.default_rng(seed)
    if "hyper_synthesis/layer_2/bias" in w:
        b = w["hyper_synthesis/layer_2/bias"].copy()
        c = b.shape[0] // 2
        b[c:] = rng.uniform(raw[0], raw[1], size=c)
        w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
        model.set_weights(w)
    last = [k[:-len("/kernel")] for k in w if k.startswith("analysis/") and k.endswith("/kernel")]
    last = "analysis/conv3" if "analysis/conv3/kernel" in w else last[-1]
    probe = data_lib.normalize_image(data_lib.synthetic_images(1, 256, 256, seed=99))
    gain = np.float32(y_std / float(model.infer_latent_rvs(probe).uq[-1].loc.std()))
    w[last + "/kernel"] = (w[last + "/kernel"] * gain).astype(np.float32)
    if last + "/bias" in w:
        w[last + "/bias"] = (w[last + "/bias"] * gain).astype(np.float32)
    model.set_weights(w)
    return model

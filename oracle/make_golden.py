"""Generates tests/golden/*.npz from the float64 oracle (run in the build container:
``python -m oracle.make_golden``).  TEST INFRASTRUCTURE.

The reference cannot be executed here (TensorFlow / TFC / TFP absent, no network) and ships no test
vectors, so these are NOT reference outputs: they freeze the oracle's own answers (inputs, weights and
float64 outputs) so that (a) the oracle cannot drift silently and (b) the GPU box, which has no
/root/reference and no need for SciPy-free re-derivation, checks the HIP kernels against committed data.
tests/golden/published_rows.json is different: it is copied DATA from the reference's published
results (results/kodak/*-detailed.json, results/all_params.csv, results/all_fpp.csv).
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from . import model_np
from . import ops_np as O
from . import transforms_np as T

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
REF = Path("/root/reference")


def ops_fixture():
    rng = np.random.default_rng(20261002)
    d = {}

    def conv_case(tag, kind, k, s, cin, cout, n, h, w):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        shp = (k, k, cout, cin) if kind == "convT" else (k, k, cin, cout)
        wk = (rng.standard_normal(shp) / np.sqrt(k * k * cin / 4)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        fn = {"conv": O.conv2d, "convT": O.conv2d_transpose, "sigdown": O.signal_conv_down, "sigup": O.signal_conv_up}[kind]
        d[f"{tag}/x"], d[f"{tag}/w"], d[f"{tag}/b"], d[f"{tag}/y"] = x, wk, b, fn(x, wk, b, s)
        d[f"{tag}/meta"] = np.array([k, s, cin, cout])

    conv_case("conv_k5s2_odd", "conv", 5, 2, 32, 24, 1, 9, 11)
    conv_case("conv_k5s2_rgb", "conv", 5, 2, 3, 16, 1, 16, 12)
    conv_case("conv_k3s1", "conv", 3, 1, 32, 32, 1, 6, 7)
    conv_case("convT_k13s8", "convT", 13, 8, 32, 24, 1, 3, 4)
    conv_case("convT_k18s16", "convT", 18, 16, 32, 3, 1, 2, 3)
    conv_case("convT_k5s2", "convT", 5, 2, 32, 40, 1, 4, 5)
    conv_case("convT_k3s1", "convT", 3, 1, 32, 16, 1, 5, 4)
    conv_case("sigdown_k9s4", "sigdown", 9, 4, 3, 16, 1, 16, 20)
    conv_case("sigup_k5s2", "sigup", 5, 2, 32, 8, 1, 4, 3)

    c = 12
    x = rng.standard_normal((1, 5, 6, c)).astype(np.float32)
    beta = (1 + rng.random(c)).astype(np.float32)
    gamma = (0.1 * np.eye(c) + 0.02 * rng.random((c, c))).astype(np.float32)
    d["gdn/x"], d["gdn/beta"], d["gdn/gamma"] = x, beta, gamma
    d["gdn/igdn1"] = O.gdn(x, beta, gamma, inverse=True)
    d["gdn/gdn1"] = O.gdn(x, beta, gamma, inverse=False)
    d["gdn/classic"] = O.gdn(x, beta, gamma, inverse=False, alpha=2, epsilon=0.5)

    # entropy: SURVEY.md 8d synthetic latents
    n, h, w, c = 2, 4, 3, 16
    mu = rng.standard_normal((n, h, w, c)).astype(np.float32)
    raw = rng.uniform(-3.0, 4.3, size=(n, h, w, c)).astype(np.float32)
    y = (mu + rng.laplace(0, 2.0, size=(n, h, w, c))).astype(np.float32)
    y_hat, bits, sym = O.scale_indexed_normal(y, mu, np.exp(raw.astype(np.float64)))
    d["normal/y"], d["normal/hyper"] = y, np.concatenate([mu, raw], -1)
    d["normal/y_hat"], d["normal/bits"], d["normal/symbols"] = y_hat, bits, sym.astype(np.int32)
    for nf in [(3, 3), (3, 3, 3)]:
        tag = "df" + "".join(map(str, nf))
        p = model_np.init_deep_factorized(c, rng, nf)
        for k in p:
            p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
        z = (3 * rng.standard_normal((n, h, w, c))).astype(np.float32)
        ms, bs, fs = model_np._prior_lists(p)
        v, zbits = O.batched_deep_factorized(z, ms, bs, fs)
        d[f"{tag}/z"], d[f"{tag}/z_hat"], d[f"{tag}/bits"] = z, v, zbits
        for k, a in p.items():
            d[f"{tag}/{k}"] = a

    # pixels
    x = (rng.integers(0, 256, size=(1, 9, 7, 3)).astype(np.float32) / np.float32(255) - np.float32(0.5))
    xh = (x + rng.normal(0, 0.04, size=x.shape)).astype(np.float32)
    d["pix/x"], d["pix/x_hat"] = x, xh
    d["pix/pixels"] = O.floats_to_pixels(xh, False)
    mses, psnrs = O.mse_psnr(O.floats_to_pixels(x, False), O.floats_to_pixels(xh, False))
    d["pix/mse"], d["pix/psnr"] = mses, psnrs
    d["pix/padded8"] = O.pad_images(x, 8).astype(np.float32)
    np.savez_compressed(OUT / "ops.npz", **d)


def small_model_fixture():
    rng = np.random.default_rng(7)
    tc = dict(analysis=dict(cls="ElicAnalysis", channels=(16, 16, 16, 32)),
              synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5),
                             activation_type="igdn", res_type="conv"))
    m = model_np.Model(tc, rd_lambda=0.02)
    p = m.init_params(seed=11)
    for k in list(p):
        leaf = k.rsplit("/", 1)[-1]
        if leaf == "bias":
            p[k] = (0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
        elif leaf == "beta":
            p[k] = (1 + 0.5 * rng.random(p[k].shape)).astype(np.float32)
    b = p["hyper_synthesis/layer_2/bias"].copy()
    b[32:] = rng.uniform(-1, 3, size=32)
    p["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    yy, xx = np.mgrid[0:50, 0:70].astype(np.float32)
    img = np.stack([128 + 60 * np.sin(xx / 9 + c) * np.cos(yy / 7 - c) + rng.normal(0, 4, size=yy.shape) for c in range(3)], -1)
    x = (np.clip(np.rint(img), 0, 255).astype(np.float32)[None] / np.float32(255) - np.float32(0.5))
    r = m.end_to_end(p, x)
    z, y = m.infer_latents(p, x)
    d = {"x": x, "config": np.array(json.dumps(tc)), "rd_lambda": np.array(0.02),
         "z": z, "y": y, "z_hat": r["z_hat"], "symbols_y": r["symbols_y"].astype(np.int32), "recon": r["recon"],
         "recon_pixels": r["recon_pixels"], "bits_z": r["bits_z"], "bits_y": r["bits_y"],
         "bpp": np.array(r["bpp"]), "mse": np.array(r["mse"]), "psnr": np.array(r["psnr"]), "rd_loss": np.array(r["rd_loss"])}
    for k, a in p.items():
        d["w/" + k] = a
    np.savez_compressed(OUT / "model_two_layer_small.npz", **d)


def bitstream_fixture():
    """A frozen rANS stream of this build's wire format (oracle/rans_np.py): 505 values over all 64 scale tables incl.
    escapes -> the exact uint16 words, for (1 segment, 64 lanes) and (3 segments, 16 lanes)."""
    from . import rans_np
    rng = np.random.default_rng(0)
    tabs = rans_np.normal_tables()
    n, E = 2, 505
    tids = rng.integers(0, 64, size=(n, E)).astype(np.int16)
    sig = np.array([0.11 * np.exp((np.log(256.0) - np.log(0.11)) / 63.0 * k) for k in range(64)])
    vals = np.rint(rng.laplace(0, 1, size=(n, E)) * sig[tids] * 1.5).astype(np.int32)
    vals[0, 16] = 20000
    vals[1, 0] = -31000
    vals[1, 504] = 32767
    d = dict(values=vals, table_ids=tids, table_sizes=np.array([len(f) for _, f in tabs], np.int32),
             table_min=np.array([lo for lo, _ in tabs], np.int32), table_freqs=np.concatenate([np.asarray(f, np.int32) for _, f in tabs]))
    for segs, lanes in ((1, 64), (3, 16)):
        eseg = -(-(-(-E // segs)) // 64) * 64
        words, lens = [], []
        for b in range(n):
            for g in range(segs):
                sl = slice(g * eseg, min(E, (g + 1) * eseg))
                w = rans_np.encode_stream(vals[b, sl], tids[b, sl], tabs, lanes)
                assert rans_np.decode_stream(w, tids[b, sl], tabs, lanes) == vals[b, sl].tolist()
                words += w
                lens.append(len(w))
        d[f"words_s{segs}"] = np.asarray(words, np.uint16)
        d[f"lens_s{segs}"] = np.asarray(lens, np.int64)
        d[f"lanes_s{segs}"] = np.asarray(lanes, np.int64)
    np.savez_compressed(OUT / "bitstream.npz", **d)


def published_rows():
    """DATA copied from the reference's published results (not source): a few per-image rows and the
    parameter / FLOP tables, used as known-answer tests of the metric definitions and the layer inventory."""
    out = {}
    for name in ["2-layer_syn", "JPEG-like_syn"]:
        rows = json.load(open(REF / "results" / "kodak" / f"{name}-detailed.json"))
        out[name] = [{k: r[k] for k in ("rd_lambda", "instance_id", "bpp", "psnr", "mse", "rd_loss")} for r in rows[:12]]
    out["params_csv"] = (REF / "results" / "all_params.csv").read_text()
    out["fpp_csv"] = (REF / "results" / "all_fpp.csv").read_text()
    agg = json.load(open(REF / "results" / "kodak" / "aggregate.json"))
    out["kodak_aggregate"] = {k: {"bpp": v["bpp"], "psnr": v["psnr"]} for k, v in agg.items()}
    (OUT / "published_rows.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    ops_fixture()
    small_model_fixture()
    bitstream_fixture()
    if REF.exists():
        published_rows()
    for f in sorted(OUT.iterdir()):
        print(f.name, f.stat().st_size)

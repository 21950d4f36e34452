#!/usr/bin/env python3
"""Achieved HBM bandwidth of the streaming (non-MFMA) kernels of the path on a Kodak-shaped batch: algorithmic bytes
/ HIP-event time.  python tools/stream_kernels.py [--batch 18] [--hw 512 768]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import entropy_coding as ec
from shallow_ntc_amd import ops
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=18)
ap.add_argument("--hw", type=int, nargs=2, default=[512, 768])
args = ap.parse_args()
dev = torch.device("cuda:0")
n, (h, w) = args.batch, args.hw
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
x = (torch.rand((n, h, w, 3), device=dev) - 0.5).contiguous()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


with torch.cuda.device(dev):
    lat = model.infer_latent_rvs(x)
    z, y = lat.uq[0].loc, lat.uq[1].loc
    prior = model._get_prior()
    z_hat, _ = prior(z)
    hyper = model._hyper_synthesis(z_hat)
    y_hat, _, sym = ops.entropy_scale_normal(y, hyper, True)
    mid = model._synthesis._up(y_hat)
    recon = model._synthesis(y_hat)
    t = model._synthesis
    rows = [
        ("scale_normal (y, mu, raw in; y_hat, symbols out)", lambda: ops.entropy_scale_normal(y, hyper, True), y.numel() * 4 * 5),
        ("scale_normal (no symbols)", lambda: ops.entropy_scale_normal(y, hyper, False), y.numel() * 4 * 4),
        ("deep factorized (z in, z_hat out)", lambda: prior(z), z.numel() * 4 * 2),
        ("dequant (symbols, mu in; y_hat out)", lambda: ops.dequant_scale_normal(sym, hyper), y.numel() * 4 * 3),
        ("two-layer tail (24 ch half-res in, 3 ch float out)", lambda: ops.two_layer_tail(mid, t._ch, t._has_res, t._act_kind, t._beta, t._gamma, t._w2, t._b2, t._k[1], t._s[1]),
         mid.numel() * 4 + recon.numel() * 4),
        ("pixels_sse (x, x_hat in; integer SSE)", lambda: ops.pixels_sse(x, recon), x.numel() * 4 * 2),
        ("to_pixels (x_hat in, u8 out)", lambda: ops.to_pixels(recon, h, w), recon.numel() * 5),
        ("pad_reflect (copy)", lambda: ops.pad_reflect(x, h + 64, w + 64) if hasattr(ops, "pad_reflect") else None, x.numel() * 4 * 2),
        ("scale_table_ids (raw in, u16 out)", lambda: ec.scale_table_ids(hyper), y.numel() * 6),
    ]
    for name, fn, nbytes in rows:
        try:
            ms = timed(fn)
        except Exception as e:   # noqa: BLE001
            print(f"{name:58s} skipped ({e})")
            continue
        print(f"{name:58s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s  ({nbytes / 1e6:.1f} MB)")

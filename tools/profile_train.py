#!/usr/bin/env python3
"""Training-step time at the reference's training shape (two_layer_syn.py: batch 8 x 256 x 256) and where it goes.
python tools/profile_train.py [--batch 8] [--hw 256 256] [--steps 10]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import ops
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
from shallow_ntc_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--hw", type=int, nargs=2, default=[256, 256])
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = configs.CONFIGS["two_layer_syn"]()
cfg["optimizer_config"] = dict(learning_rate=1e-4, global_clipnorm=1.0)
model = Model(device=dev, **cfg)
tr = Trainer(model)
n, (h, w) = args.batch, args.hw
x = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=3))).to(dev)
print(f"variables: {tr.store.total / 1e6:.2f} M floats, buckets {tr._bucket_slices()}")
for _ in range(2):
    m = tr.train_step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    m = tr.train_step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(f"train step: {dt * 1e3:.2f} ms for {n} x {h}x{w} -> {n / dt:.1f} images/s, {n * h * w / dt / 1e6:.2f} Mpixel/s; loss {m['rd_loss']:.4f}")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


with torch.cuda.device(dev):
    t_fb = timed(lambda: tr.loss_and_grads(x, 0.08))
    t_fwd = timed(lambda: (tr.analysis.fwd(x)))
    t_refresh = timed(tr._refresh)
    t_adam = timed(lambda: ops.adam_step(tr.store.param, tr.store.grad, tr.store.m, tr.store.v, 0.0, 1))
    y, k_a = tr.analysis.fwd(x)
    g = torch.randn_like(y)
    t_abwd = timed(lambda: tr.analysis.bwd(k_a, g, need_dx=False))
print(f"forward+backward {t_fb:.2f} ms (analysis fwd {t_fwd:.2f}, analysis bwd {t_abwd:.2f}); plan refresh {t_refresh:.2f} ms; adam {t_adam:.3f} ms")
for kind, k, s, cin, cout, hh in (("conv", 5, 2, 192, 192, h // 2), ("conv", 3, 1, 96, 96, h // 2), ("conv", 1, 1, 192, 96, h // 2),
                                  ("conv", 1, 1, 96, 192, h // 2), ("conv", 5, 2, 3, 192, h), ("convT", 13, 8, 320, 24, h // 16)):
    xi = torch.randn((n, hh, hh * w // h, cin), device=dev)
    ho = hh * s if kind == "convT" else -(-hh // s)
    go = torch.randn((n, ho, ho * w // h, cout), device=dev)
    dw = torch.empty((k, k, cin, cout) if kind == "conv" else (k, k, cout, cin), device=dev)
    ms = timed(lambda: ops.conv_wgrad(kind, k, s, cin, cout, xi, go, dw))
    px = n * (ho if kind == "conv" else hh) * (ho if kind == "conv" else hh) * w // h
    fl = 2.0 * px * k * k * cin * cout
    print(f"  wgrad {kind:5s} k{k} s{s} {cin:3d}->{cout:3d} @{hh}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")

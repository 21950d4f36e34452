#!/usr/bin/env python3
"""SGA step time (SURVEY.md row a21 hot loop).  python tools/profile_sga.py [--batch 1] [--hw 512 768] [--steps 20]"""
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
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--hw", type=int, nargs=2, default=[512, 768])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--config", default="two_layer_syn")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = {**configs.CONFIGS[args.config](rd_lambda=0.02), **configs.itinf()}
model = Model(device=dev, **cfg)
x = (torch.rand((args.batch, *args.hw, 3), device=dev) - 0.5).contiguous()
model.initialize_itinf(x)
for _ in range(3):
    model.itinf_train_step(x, seed=1, fetch=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    model.itinf_train_step(x, seed=1, fetch=False)        # as the loop driver runs it: no host synchronisation per step
torch.cuda.synchronize()
model.itinf_last_metrics()
dt = (time.perf_counter() - t0) / args.steps
print(f"SGA step: {dt * 1e3:.3f} ms  ({args.batch} x {args.hw[0]}x{args.hw[1]}) -> 3000 steps = {3000 * dt:.1f} s")
ops.PROFILE = []
model.itinf_train_step(x, seed=1)
torch.cuda.synchronize()
tot = 0
for e in ops.PROFILE:
    ms = e["e0"].elapsed_time(e["e1"])
    tot += ms
    print(f"  {e['kind']:6s} k{e['k']} s{e['s']} {e['cin']:4d}->{e['cout']:4d} in {e['n']}x{e['h']}x{e['w']:<4d} v{e['variant']} {ms:8.4f} ms {e['flops'] / ms / 1e9:7.1f} TF")
print(f"  conv total {tot:.3f} ms")
# host (launch) time of a step vs time until the GPU is done: is the loop launch-bound?
host = []
for _ in range(10):
    t0 = time.perf_counter()
    model.itinf_train_step(x)
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f"host time per step (no sync inside the loop body would hide it): {1e3 * sorted(host)[5]:.3f} ms")

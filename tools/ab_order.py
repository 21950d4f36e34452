#!/usr/bin/env python3
"""A/B of the fp32 stream-K unit order on one layer: row strip outermost against column tile outermost (the COLM twin of the
128 x 128 instance), interleaved bursts in one process, bit equality checked.  Defaults: the 480 -> 640 hyper-synthesis layer at
the Kodak batch.  python tools/ab_order.py [--n 18] [--hw 32 48] [--cin 480] [--cout 640]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=18)
ap.add_argument("--hw", type=int, nargs=2, default=[32, 48])
ap.add_argument("--cin", type=int, default=480)
ap.add_argument("--cout", type=int, default=640)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
h, w = args.hw
x = torch.randn((args.n, h, w, args.cin), device=dev, generator=g)
wk = torch.randn((3, 3, args.cout, args.cin), device=dev, generator=g) * 0.02
b = torch.randn((args.cout,), device=dev, generator=g)
plans = {}
for name, colm in (("strip-major", False), ("column-major", True), ("rule", None)):
    p = ops.ConvPlan("convT", wk, b, 1, "relu")
    p.set_tile(9)
    p.set_stream_k(True, colm=colm)
    plans[name] = p
y = {k: torch.empty((args.n, h, w, args.cout), device=dev) for k in plans}
flops = plans["rule"].flops(args.n, h, w)
times = {k: [] for k in plans}
for rep in range(args.reps + 2):
    for k, p in plans.items():
        p(x, out=y[k])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            p(x, out=y[k])
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            times[k].append(e0.elapsed_time(e1) / 6)
ops.check_conv_status()
for k in plans:
    ms = float(np.median(times[k]))
    print("%-13s %.4f ms  %.1f TFLOP/s  launch %s  bit-identical to strip-major: %s"
          % (k, ms, flops / ms / 1e9, plans[k].launch_info(args.n, h, w), torch.equal(y[k], y["strip-major"])))

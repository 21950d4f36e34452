#!/usr/bin/env python3
"""The RGB first layer on its own kernel (csrc/rgb_conv.hip) against the row-packed gather-GEMM plan + zero-padding pass it
replaces: bit-identity and time (bursts of launches between HIP events, interleaved rounds in one process), with the two roofs of
the layer next to it: K = 80 MFMA issue at 157.3 TFLOP/s and the output store stream at 8 TB/s.
python tools/rgb_conv_block.py [--shapes 18x512x768,6x768x512,64x256x256,1x512x768]"""
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
ap.add_argument("--shapes", default="18x512x768,6x768x512,64x256x256,1x512x768,5x1200x1200")
ap.add_argument("--cout", type=int, default=192)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
w = torch.randn((5, 5, 3, args.cout), device=dev, generator=g) * 0.1
b = torch.randn((args.cout,), device=dev, generator=g)
rp = ops.RowPackedConv(w, b, 2)
plan = ops.RgbConvPlan(w, b, 2)


def burst(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


print(f"{'shape':>16s} {'row-packed + pad':>18s} {'own kernel':>12s} {'TFLOP/s':>8s} {'out TB/s':>9s} {'MFMA roof':>10s} {'store roof':>11s}  same bits")
for spec in args.shapes.split(","):
    n, h, wd = (int(v) for v in spec.split("x"))
    x = (torch.rand((n, h, wd, 3), device=dev, generator=g) - 0.5).contiguous()
    same = torch.equal(rp(x), plan(x))
    ta, tb = [], []
    for _ in range(args.rounds):
        ta.append(burst(lambda: rp(x)))
        tb.append(burst(lambda: plan(x)))
    a, bb = float(np.median(ta)), float(np.median(tb))
    fl = plan.flops(n, h, wd)
    out_bytes = n * (-(-h // 2)) * (-(-wd // 2)) * args.cout * 4
    print(f"{spec:>16s} {a:15.4f} ms {bb:9.4f} ms {fl / bb / 1e9:8.1f} {out_bytes / bb / 1e9:9.2f} {fl * 80 / 75 / 157.3e9:7.4f} ms {out_bytes / 8e9:8.4f} ms  {same}")

#!/usr/bin/env python3
"""ResidualBlock tail (conv3x3 96 -> 96 relu, conv1x1 96 -> 192 + skip): two launches against the fused one, same inputs.

    python tools/fused_tail.py [--n 18 --hw 256 384 --reps 10]
With the DIAG library (SNTC_LIB=...libsntc_hip_diag.so) SNTC_GG_DBG=128 drops the fused launch's second contraction: what is
left is the 3x3 at the fused instance's occupancy."""
import argparse, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd import _capi as capi
from shallow_ntc_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=18)
ap.add_argument("--hw", type=int, nargs=2, default=[256, 384])
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
h, w = args.hw
x = torch.randn((args.n, h, w, 96), device=dev, generator=g)
res = torch.randn((args.n, h, w, 192), device=dev, generator=g)
first = ops.ConvPlan("conv", torch.randn((3, 3, 96, 96), device=dev, generator=g) * 0.05, torch.randn((96,), device=dev, generator=g), 1, "relu")
second = ops.ConvPlan("conv", torch.randn((1, 1, 96, 192), device=dev, generator=g) * 0.1, torch.randn((192,), device=dev, generator=g), 1, None,
                      capi.PRO_NONE, capi.EPI_ADD)


def timed(fn, burst=6):
    """Median over bursts of back-to-back launches (a lone launch from an idle stream carries ~0.1 ms of host latency)."""
    ts = []
    for rep in range(args.reps + 3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(burst):
            fn()
        e1.record(); torch.cuda.synchronize()
        if rep >= 3:
            ts.append(e0.elapsed_time(e1) / burst)
    return float(np.median(ts))


flops = first.flops(args.n, h, w) + second.flops(args.n, h, w)
t3 = timed(lambda: first(x))
t2 = timed(lambda: second(first(x), res=res))
t1 = timed(lambda: first.fused(second, x, res=res))
print(f"{args.n}x{h}x{w}: 3x3 alone {t3:.4f} ms | two launches {t2:.4f} ms ({flops / t2 / 1e9:.1f} TFLOP/s) | fused {t1:.4f} ms ({flops / t1 / 1e9:.1f} TFLOP/s)")

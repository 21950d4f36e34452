#!/usr/bin/env python3
"""Run ONE convolution layer repeatedly (for rocprofv3 --pmc / --kernel-trace passes and variant A/B rounds).

    python tools/one_layer.py --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 [--variant V] [--static]
        [--reps 30] [--ab 2,3,4]      # --ab: interleaved rounds of several tile variants in one process
"""
import argparse
import sys
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
ap.add_argument("--kind", default="convT")
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--s", type=int, default=1)
ap.add_argument("--cin", type=int, default=480)
ap.add_argument("--cout", type=int, default=640)
ap.add_argument("--n", type=int, default=18)
ap.add_argument("--hw", type=int, nargs=2, default=[32, 48])
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--static", action="store_true")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--ab", default="")
ap.add_argument("--epi", action="store_true", help="residual-add epilogue")
ap.add_argument("--dma", type=int, default=-1, help="1 / 0: force the direct-to-LDS / register stage path (default: library default)")
ap.add_argument("--bf16x3", action="store_true", help="split-precision experiment: also run the bf16 x 3 plan and compare with fp32")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
h, w = args.hw
x = torch.randn((args.n, h, w, args.cin), device=dev, generator=g)
wshape = (args.k, args.k, args.cout, args.cin) if args.kind == "convT" else (args.k, args.k, args.cin, args.cout)
wk = torch.randn(wshape, device=dev, generator=g) * 0.05
b = torch.randn((args.cout,), device=dev, generator=g)


def make(variant):
    p = ops.ConvPlan(args.kind, wk, b, args.s, None, capi.PRO_NONE, capi.EPI_ADD if args.epi else capi.EPI_STORE)
    if variant:
        p.set_tile(variant)
    if args.static or args.dma >= 0:
        p.set_stream_k(not args.static, dma=None if args.dma < 0 else bool(args.dma))
    return p


variants = [int(v) for v in args.ab.split(",")] if args.ab else [args.variant]
plans = {v: make(v) for v in variants}
ho, wo = plans[variants[0]].out_hw(h, w)
res = torch.randn((args.n, ho, wo, args.cout), device=dev, generator=g) if args.epi else None
y = torch.empty((args.n, ho, wo, args.cout), device=dev)
flops = plans[variants[0]].flops(args.n, h, w)
times = {v: [] for v in variants}
BURST = 6     # launches per measurement, back to back: a lone launch from an idle stream is timed with the host's own latency in it


def timed_burst(fn):
    fn()                                   # the burst's first launch hides the host latency of the rest
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(BURST):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / BURST


for rep in range(args.reps + 2):
    for v in variants:
        ms = timed_burst(lambda: plans[v](x, res=res, out=y))
        if rep >= 2:
            times[v].append(ms)
if args.bf16x3:
    ref = plans[variants[0]](x, res=res).clone()
    y3 = torch.empty_like(y)
    p3 = ops.ConvPlan(args.kind, wk, b, args.s, None, capi.PRO_NONE, capi.EPI_ADD if args.epi else capi.EPI_STORE, bf16x3=True)
    for v3 in (2, 4):
        p3.set_tile(v3)
        ts = [timed_burst(lambda: p3(x, res=res, out=y3)) for _ in range(args.reps)]
        err = float((y3 - ref).abs().max() / ref.abs().max())
        vv, nb = p3.launch_info(args.n, h, w)
        print(f"bf16x3 (split while staging) variant {vv} blocks {nb}: median {np.median(ts):.4f} ms  {flops / np.median(ts) / 1e9:.1f} TFLOP/s-equivalent; "
              f"max |bf16x3 - fp32| / max |fp32| = {err:.2e}")
    ps = ops.ConvPlan(args.kind, wk, b, args.s, None, capi.PRO_NONE, capi.EPI_ADD if args.epi else capi.EPI_STORE, bf16x3="presplit")
    xs = ops.split3(x)
    for v3 in (11, 12, 13):
        ps.set_tile(v3)
        for static, halo in ((False, True), (True, True), (False, False), (True, False)):
            ps.set_stream_k(not static, force=not static, halo=halo)
            ts = [timed_burst(lambda: ps(xs, res=res, out=y3)) for _ in range(args.reps)]
            err = float((y3 - ref).abs().max() / ref.abs().max())
            vv, nb = ps.launch_info(args.n, h, w)
            print(f"bf16x3 pre-split variant {vv} {'static  ' if static else 'stream-K'} {'patch  ' if halo else 'per-tap'} blocks {nb}: median {np.median(ts):.4f} ms  "
                  f"{flops / np.median(ts) / 1e9:.1f} TFLOP/s-equivalent; max |bf16x3 - fp32| / max |fp32| = {err:.2e}")
    ts = [timed_burst(lambda: ops.split3(x)) for _ in range(args.reps)]
    print(f"split3 of the input: median {np.median(ts):.4f} ms ({x.numel() * 10 / np.median(ts) / 1e6:.0f} GB/s)")
for v in variants:
    t = np.array(times[v])
    vv, nb = plans[v].launch_info(args.n, h, w)
    print(f"{args.kind} k{args.k} s{args.s} {args.cin}->{args.cout} {args.n}x{h}x{w} variant {vv} blocks {nb}: "
          f"median {np.median(t):.4f} ms min {t.min():.4f} ms  {flops / np.median(t) / 1e9:.1f} TFLOP/s (min-time {flops / t.min() / 1e9:.1f})")

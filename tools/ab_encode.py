#!/usr/bin/env python3
"""A/B of the ENCODE region exactly as bench.py runs it (the Kodak-shaped set's two batch shapes on two streams), with a library
switch flipped between two models that share their weights: interleaved rounds in one process, so that the box's clock and
its neighbours cancel.  python tools/ab_encode.py [--flag RGB_FIRST_LAYER] [--autotune]"""
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
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--flag", default="RGB_FIRST_LAYER")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--autotune", action="store_true")
ap.add_argument("--one-stream", action="store_true")
ap.add_argument("--rgb-wgs", type=int, default=0, help="cap / raise the persistent workgroups of the RGB first-layer kernel (0: one per CU)")
args = ap.parse_args()
dev = torch.device("cuda:0")
side = ops.side_streams(3, dev)
models = {}
shared = [(torch.rand((n, h, w, 3), device=dev) - 0.5).contiguous() for n, h, w in ((6, 768, 512), (18, 512, 768))]
for val in (True, False):
    setattr(ops, args.flag, val)
    m = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.005))
    xs = shared
    for x in xs:
        m.encode(x, check=False)          # builds the plans under this value of the switch
    first = m._analysis._graph.layers[0].plan
    if args.rgb_wgs and isinstance(first, ops.RgbConvPlan):
        first.set_workgroups(args.rgb_wgs)
    models[val] = (m, xs)
setattr(ops, args.flag, True)


def step(m, xs):
    if args.one_stream:
        return [m.encode(x, check=False) for x in xs]
    cur = torch.cuda.current_stream()
    outs = []
    for i, x in enumerate(xs):
        st = side[i % 2]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(m.encode(x, check=False))
    for st in side[:2]:
        cur.wait_stream(st)
    return outs


if args.autotune:
    for val, (m, xs) in models.items():
        with ops.autotune():
            for x in xs:
                m.encode(x, check=False)
a = step(*models[True]); b = step(*models[False])
torch.cuda.synchronize()
same = all(torch.equal(p, q) for u, v in zip(a, b) for p, q in zip(u, v))
ts = {True: [], False: []}
for r in range(args.rounds):
    for val in (True, False):
        m, xs = models[val]
        step(m, xs); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step(m, xs)
        e1.record(); torch.cuda.synchronize()
        ts[val].append(e0.elapsed_time(e1) / args.steps)
ops.check_conv_status()
for val in (True, False):
    print(f"{args.flag} = {val!s:5s}: encode of the 24-image set {np.median(ts[val]):.3f} ms per step (rounds: {' '.join(f'{t:.2f}' for t in ts[val])})")
print("same outputs:", same)

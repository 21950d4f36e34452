#!/usr/bin/env python3
"""The two-stream Kodak decode step (bench.py's `value` region) under ops.tune_step: every (tile, schedule) candidate of its
convolution launches, and the synthesis launch's workgroup cap, judged by the STEP's own clock (the launches overlap on the
device, so a schedule measured alone can be the wrong one).  Prints the step before / after and what was chosen.
python tools/tune_decode.py [--reps 15]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from shallow_ntc_amd import ops  # noqa: E402
from shallow_ntc_amd.mshyper import configs  # noqa: E402
from shallow_ntc_amd.mshyper.models import Model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=15)
args = ap.parse_args()
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.005))
g = torch.Generator(device=dev)
g.manual_seed(99)
codes = []
for n, (h, w) in ((6, (768, 512)), (18, (512, 768))):          # smallest batch first, as bench.py feeds them
    z_hat = torch.round(3.0 * torch.randn((n, h // 64, w // 64, 320), device=dev, generator=g)).contiguous()
    u = torch.rand((n, h // 16, w // 16, 320), device=dev, generator=g) - 0.5
    sym = torch.round(-2.0 * torch.sign(u) * torch.log1p(-2.0 * u.abs())).to(torch.int32).contiguous()
    codes.append((z_hat, sym, (h, w)))
side = [torch.cuda.Stream(device=dev) for _ in codes]


def step():
    cur = torch.cuda.current_stream()
    outs = []
    for st, (z_hat, sym, hw) in zip(side, codes):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(model.decode(z_hat, sym, hw, check=False))
    for st in side:
        cur.wait_stream(st)
    return outs


def clock(reps):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


want = [o.clone() for o in step()]
for _ in range(5):
    step()
print(f"step, cost-model schedules: {clock(40):.4f} ms", flush=True)
syn = model._synthesis._syn
if syn is not None:
    for wg in (0, 224, 240, 248, 192, 128):
        syn.set_workgroups(wg)
        print(f"  synthesis workgroups {wg or 'one per CU'}: {clock(30):.4f} ms", flush=True)
    syn.set_workgroups(0)
log = []
before, after = ops.tune_step(step, reps=args.reps, log=log)
for row in log:
    print("  ", row, flush=True)
print(f"tune_step: {before:.4f} -> {after:.4f} ms; re-measured {clock(40):.4f} ms", flush=True)
assert all(torch.equal(a, b) for a, b in zip(step(), want)), "tuned launches changed the pixels"
ops.check_conv_status()

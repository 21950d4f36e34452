#!/usr/bin/env python3
"""The two-stream Kodak encode step (bench.py's `encode` region) under ops.tune_step, after the per-layer autotune pass bench.py
runs: does the step's own clock find anything the device-idle measurements did not?  python tools/tune_encode.py"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft  # noqa: E402
graft.load_package()
from shallow_ntc_amd import ops  # noqa: E402
from shallow_ntc_amd.common import data_lib  # noqa: E402
from shallow_ntc_amd.mshyper import configs  # noqa: E402
from shallow_ntc_amd.mshyper.models import Model  # noqa: E402

dev = torch.device("cuda:0")
model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.005))
batches = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=n))).to(dev) for n, (h, w) in ((6, (768, 512)), (18, (512, 768)))]
side = [torch.cuda.Stream(device=dev) for _ in batches]


def step():
    cur = torch.cuda.current_stream()
    outs = []
    for st, x in zip(side, batches):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(model.encode(x, check=False))
    for st in side:
        cur.wait_stream(st)
    return outs


def clock(reps=7):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


want = [o[1].clone() for o in step()]
print(f"cost model: {clock():.3f} ms", flush=True)
with ops.autotune():
    for x in batches:
        model.encode(x, check=False)
torch.cuda.synchronize()
print(f"per-layer autotune (device idle): {clock():.3f} ms", flush=True)
log = []
t0 = time.perf_counter()
before, after = ops.tune_step(step, reps=5, burst=1, passes=1, max_launches=24, log=log)
print(f"tune_step ({time.perf_counter() - t0:.1f} s): {before:.3f} -> {after:.3f} ms; re-measured {clock():.3f} ms", flush=True)
for r in log:
    if r["chosen"]:
        print("  ", r, flush=True)
assert all(torch.equal(a[1], b) for a, b in zip(step(), want))
ops.check_conv_status()

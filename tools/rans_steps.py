#!/usr/bin/env python3
"""rANS decode launch time against the number of steps per stream (HIP events around the sntc_rans_decode call alone): the
per-step cost of a lone wave and the launch's fixed part (tables into LDS, first ring fills).  Scale tables, 64 lanes, 36 streams.
python tools/rans_steps.py [--no-lut]"""
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
from shallow_ntc_amd import entropy_coding as ec

ap = argparse.ArgumentParser()
ap.add_argument("--no-lut", action="store_true")
ap.add_argument("--streams", type=int, default=36)
ap.add_argument("--sigma-index", type=int, default=-1, help="all elements on this scale table (default: uniformly random tables)")
args = ap.parse_args()
if args.no_lut:
    ec.USE_START_TABLES = False
dev = torch.device("cuda:0")
dt = ec.DeviceTables(ec.normal_tables(), dev)
sig = np.array([0.11 * np.exp(ec.SCALE_FACTOR * k) for k in range(64)])
rng = np.random.default_rng(0)
spans = []
orig = capi.call


def timed_call(name, *a):
    if name != "sntc_rans_decode":
        return orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(name, *a)
    e1.record()
    spans.append((e0, e1))
    return out


capi.call = timed_call
ec.capi.call = timed_call
for steps in (60, 480, 960, 1920, 3840):
    E = steps * 64
    tids = (rng.integers(0, 64, size=(args.streams, E, 1)) if args.sigma_index < 0 else np.full((args.streams, E, 1), args.sigma_index)).astype(np.int16)
    vals = np.rint(rng.standard_normal((args.streams, E, 1)) * sig[tids]).astype(np.int32)
    v, t = torch.from_numpy(vals).to(dev), torch.from_numpy(tids).to(dev)
    payload, lens = ec.rans_encode(v, t, dt, 1, 64)
    for _ in range(2):
        back = ec.rans_decode(payload, lens, t, tuple(v.shape), dt, 1, 64)
    assert torch.equal(back, v)
    spans.clear()
    for _ in range(7):
        ec.rans_decode(payload, lens, t, tuple(v.shape), dt, 1, 64)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in spans]))
    print("steps %5d: %.3f ms per launch (%d streams, %.1f bits per symbol)" % (steps, ms, args.streams, 16.0 * float(lens.sum()) / (args.streams * E)))

#!/usr/bin/env python3
"""Where the time of Model.decompress goes, on bench.py's own bitstream workload (the Kodak-shaped set, one blob per batch
shape, hyper-synthesis bias set so that the scales are in a coding range): HIP events around the stages of every call, the
host's wall clock around the whole call, and what is left between them.  python tools/decompress_phases.py [--reps 10]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import entropy_coding as ec
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--no-lut", action="store_true")
ap.add_argument("--one-by-one", action="store_true", help="one decompress() per blob instead of decompress_many()")
ap.add_argument("--small-first", action="store_true", help="hand decompress_many the small blob first (bench.py's order)")
ap.add_argument("--two-phase", action="store_true", help="A/B: round 4's schedule (all hyper-syntheses, then all latents side by side, then the syntheses)")
args = ap.parse_args()
if args.no_lut:
    ec.USE_START_TABLES = False
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
w = dict(model.get_weights())
b = w["hyper_synthesis/layer_2/bias"].copy()
b[320:] = np.random.default_rng(0).uniform(-1.0, 2.5, size=320)
w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
model.set_weights(w)
batches = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(n, h, ww, seed=s))).to(dev)
           for n, h, ww, s in ((18, 512, 768, 1), (6, 768, 512, 2))]
blobs = [model.compress(x) for x in batches]
if args.small_first:
    blobs = blobs[::-1]
if args.two_phase:
    ec.PIPELINE_BLOBS = False
print("blob bytes", [len(b) for b in blobs], "bpp %.4f" % (8.0 * sum(len(b) for b in blobs) / (24 * 512 * 768)))

marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        marks.append((label, e0, e1))
        return out

    setattr(obj, name, inner)


wrap(ec, "rans_decode", "rans_decode")
wrap(model, "_hyper_synthesis", "hyper_synthesis")
wrap(model, "_pixels", "synthesis+pixels")
run = (lambda: [model.decompress(b) for b in blobs]) if args.one_by_one else (lambda: model.decompress_many(blobs))
for _ in range(3):
    run()
torch.cuda.synchronize()
rows = []
for _ in range(args.reps):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    per = {}
    order = []
    for i, (label, e0, e1) in enumerate(marks):
        key = "%d:%s" % (i, label)
        per[key] = e0.elapsed_time(e1)
        order.append(key)
    per["wall"] = wall
    rows.append(per)
keys = list(rows[0].keys())
ref = [torch.cat([p.reshape(-1) for p in [model.decompress(b) for b in blobs]]), torch.cat([p.reshape(-1) for p in model.decompress_many(blobs)])]
assert torch.equal(ref[0], ref[1])
print("median over %d passes of both blobs (ms):" % args.reps)
tot = 0.0
for k in keys:
    v = float(np.median([r[k] for r in rows]))
    if k not in ("span", "wall"):
        tot += v
    print("  %-24s %.3f" % (k, v))
print("  %-24s %.3f  (stages that ran side by side count twice; else host work, copies, parsing, the final read-back)"
      % ("wall - sum of stages", float(np.median([r["wall"] for r in rows])) - tot))

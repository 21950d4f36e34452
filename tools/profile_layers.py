#!/usr/bin/env python3
"""Per-launch table of the convolution kernels in the decode and encode regions (HIP events on the
launch stream).  python tools/profile_layers.py [--batch 18] [--hw 512 768] [--variant V]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import _capi, ops
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=18)
ap.add_argument("--hw", type=int, nargs=2, default=[512, 768])
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--config", default="two_layer_syn")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--decode-only", action="store_true")
ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"])
ap.add_argument("--autotune", action="store_true", help="measure every plan's (tile, schedule) candidates first (ops.autotune), as bench.py does for its encoder-side regions")
args = ap.parse_args()
dev = torch.device("cuda:0")
ops.FORCE_TILE = args.variant
model = Model(device=dev, precision=args.precision, **configs.CONFIGS[args.config]())
n, (h, w) = args.batch, args.hw
x = (torch.rand((n, h, w, 3), device=dev) - 0.5).contiguous()


def run(fn, label):
    if args.autotune:
        with ops.autotune():
            fn()
    fn()
    torch.cuda.synchronize()
    acc = {}
    for _ in range(args.reps):
        ops.PROFILE = []
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); fn(); t1.record()
        torch.cuda.synchronize()
        for i, e in enumerate(ops.PROFILE):
            a = acc.setdefault(i, dict(e, samples=[]))
            a["samples"].append(e["e0"].elapsed_time(e["e1"]))
        total = t0.elapsed_time(t1)
        ops.PROFILE = None
    print(f"== {label}: {total:.3f} ms wall, {n * h * w / total / 1e3:.1f} Mpx/s")
    tot_ms = tot_fl = 0
    for i, a in acc.items():
        a["ms"] = float(np.median(a["samples"]))          # median: a rep now and then carries a one-off stall
        tf = a["flops"] / a["ms"] / 1e9
        tot_ms += a["ms"]; tot_fl += a["flops"]
        print(f"{i:3d} {a['kind']:7s} k{a['k']} s{a['s']} {a['cin']:4d}->{a['cout']:4d} in {a['n']}x{a['h']}x{a['w']:<4d} v{a['variant']} "
              f"blk {a['nblocks']:6d} {a['ms']:8.4f} ms {tf:7.1f} TF {a['flops']/1e9:9.2f} GF")
    print(f"   conv total {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")


z_hat, sym, _, _ = model.encode(x)
run(lambda: model.decode(z_hat, sym, (h, w), check=False), "decode")
if not args.decode_only:
    run(lambda: model.encode(x, check=False), "encode")

#!/usr/bin/env python3
"""Time the rANS stages of Model.compress / decompress (HIP events on the launch stream) and report the
real-vs-estimated rate.  python tools/profile_bitstream.py [--batch 18] [--hw 512 768] [--segments 2]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import entropy_coding as ec
from shallow_ntc_amd import ops
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=18)
ap.add_argument("--hw", type=int, nargs=2, default=[512, 768])
ap.add_argument("--segments", type=int, default=2)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
ec.ELEMS_PER_SEGMENT = -(-(args.hw[0] // 16) * (args.hw[1] // 16) * 320 // args.segments)
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
n, (h, w) = args.batch, args.hw
x = data_lib.synthetic_images(n, h, w, seed=1234) if hasattr(data_lib, "synthetic_images") else None
if x is None:
    x = np.random.default_rng(1234).integers(0, 256, (n, h, w, 3)).astype(np.uint8)
codec = model._get_codec()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), out


with torch.cuda.device(dev):
    xf = model._as_device_images(x)
    lat = model.infer_latent_rvs(xf)
    z, y = lat.uq[0].loc, lat.uq[1].loc
    zi = ec.round_to_int(z)
    hyper = model._hyper_synthesis(ec.int_to_float(zi))
    _, bits_y, sym = ops.entropy_scale_normal(y, hyper, want_symbols=True)
    ztid, ytid = ec.channel_table_ids(z.shape, dev), ec.scale_table_ids(hyper)
    t_ze, (zp, zl) = timed(lambda: ec.rans_encode(zi, ztid, codec.z_tables, args.segments))
    t_ye, (yp, yl) = timed(lambda: ec.rans_encode(sym, ytid, codec.y_tables, args.segments))
    t_zd, zi2 = timed(lambda: ec.rans_decode(zp, zl, ztid, tuple(z.shape), codec.z_tables, args.segments))
    t_yd, sy2 = timed(lambda: ec.rans_decode(yp, yl, ytid, tuple(y.shape), codec.y_tables, args.segments))
    assert torch.equal(zi2, zi) and torch.equal(sy2, sym)
    blob = model.compress(x)
    px = model.decompress(blob)
    m = model.evaluate_batched(x) if hasattr(model, "evaluate_batched") else None
mpx = n * h * w / 1e6
print(f"segments {args.segments}: streams z {len(zl)} y {len(yl)}; encode z {t_ze:.3f} ms y {t_ye:.3f} ms (incl. length readback + compaction); "
      f"decode z {t_zd:.3f} ms y {t_yd:.3f} ms for {mpx:.2f} Mpixel")
print(f"bitstream {8 * len(blob) / (n * h * w):.4f} bpp real; y payload {16 * int(yl.sum()) / (n * h * w):.4f} bpp vs estimate "
      f"{float(bits_y.sum()) / (n * h * w):.4f} bpp")

#!/usr/bin/env python3
"""Per-launch times of ONE image through Model.evaluate()'s pass (batch 1, serial), next to the same layers' share of a
batch-18 pass: where the one-image-at-a-time flow loses against the batched one."""
import argparse, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd import ops
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--batch", type=int, default=18)
args = ap.parse_args()
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
model._quality_metrics = False


def table(n):
    x = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(n, 512, 768, seed=3))).to(dev)
    for _ in range(2):
        model._launch_frame(x)
    torch.cuda.synchronize()
    rows = None
    for _ in range(args.reps):
        ops.PROFILE = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model._launch_frame(x)
        e1.record()
        torch.cuda.synchronize()
        cur = [(e, e["e0"].elapsed_time(e["e1"])) for e in ops.PROFILE]
        tot = e0.elapsed_time(e1)
        ops.PROFILE = None
        if rows is None or tot < rows[1]:
            rows = (cur, tot)
    return rows


(b1, t1), (bn, tn) = table(1), table(args.batch)
print(f"{'layer':34s} {'v':>2s} {'blocks':>6s} {'b1 us':>8s} {'b1 TF':>7s} | {'v':>2s} {'blocks':>6s} {'bN us/img':>9s} {'bN TF':>7s}  ratio")
s1 = sn = 0.0
for (e, ms), (f, msn) in zip(b1, bn):
    name = f"{e['kind']} {e['k']}x{e['k']} s{e['s']} {e['cin']}->{e['cout']} {e['h']}x{e['w']}"
    us1, usn = ms * 1e3, msn * 1e3 / args.batch
    s1 += us1; sn += usn
    print(f"{name:34s} {e['variant']:2d} {e['nblocks']:6d} {us1:8.1f} {e['flops'] / ms / 1e9:7.1f} | {f['variant']:2d} {f['nblocks']:6d} {usn:9.1f} {f['flops'] / msn / 1e9:7.1f}  {us1 / usn:5.2f}")
print(f"conv launches: b1 {s1:.0f} us, batched {sn:.0f} us per image; whole pass: b1 {t1 * 1e3:.0f} us, batched {tn * 1e3 / args.batch:.0f} us per image")

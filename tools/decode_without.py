#!/usr/bin/env python3
"""What the small launches of the two-stream decode step are worth, bounded by REMOVING them (their outputs cached from a full pass;
timing only): the dequantisation (y_hat = symbols + mu) and the output layer (5x5/2 12 -> 3 + crop + uint8).  An upper bound on what
fusing either into its neighbour (VERDICT r5 item 3 b / c) could return.  Bursts of four steps between HIP events, interleaved rounds.
python tools/decode_without.py"""
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

dev = torch.device("cuda:0")
side = ops.side_streams(3, dev)
m = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.005))
g = torch.Generator(device=dev)
g.manual_seed(99)
codes = []
for n, h, w in ((6, 768, 512), (18, 512, 768)):
    z_hat = torch.round(3.0 * torch.randn((n, h // 64, w // 64, 320), device=dev, generator=g)).contiguous()
    u = torch.rand((n, h // 16, w // 16, 320), device=dev, generator=g) - 0.5
    sym = torch.round(-2.0 * torch.sign(u) * torch.log1p(-2.0 * u.abs())).to(torch.int32).contiguous()
    codes.append(dict(z=z_hat, s=sym, hw=(h, w)))
syn = m._synthesis
for c in codes:                                    # the full pass once: caches, plans, reference pixels
    hyper = m._hyper_synthesis(c["z"])
    c["y_hat"] = ops.dequant_scale_normal(c["s"], hyper)
    c["hid"] = syn._syn(c["y_hat"])
    c["px"] = syn.pixels_from_hidden(c["hid"], *c["hw"])[0]
    assert torch.equal(c["px"], m.decode(c["z"], c["s"], c["hw"]))


def one(c, no_dequant, no_tail):
    hyper = m._hyper_synthesis(c["z"])
    y_hat = c["y_hat"] if no_dequant else ops.dequant_scale_normal(c["s"], hyper)
    hid = syn._syn(y_hat)
    return hid if no_tail else syn.pixels_from_hidden(hid, *c["hw"])[0]


def step(no_dequant, no_tail):
    cur = torch.cuda.current_stream()
    for i, c in enumerate(codes):
        st = side[i % 2]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            one(c, no_dequant, no_tail)
    for st in side[:2]:
        cur.wait_stream(st)


variants = {"full step": (False, False), "without the dequantisation launches": (True, False), "without the output-layer launches": (False, True),
            "without both": (True, True)}
ts = {k: [] for k in variants}
for _ in range(10):
    step(False, False)
torch.cuda.synchronize()
for r in range(9):
    for name, (nd, nt) in variants.items():
        step(nd, nt); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            step(nd, nt)
        e1.record(); torch.cuda.synchronize()
        ts[name].append(e0.elapsed_time(e1) / 4)
ops.check_conv_status()
base = float(np.median(ts["full step"]))
for name in variants:
    t = float(np.median(ts[name]))
    print(f"{name:40s} {t:.3f} ms per step ({t - base:+.3f})")

#!/usr/bin/env python3
"""Where the encoder's WALL time goes, node by node: every node of the analysis transform's graph (first layer, ResidualBlocks,
strided layers, the two attention blocks) and the hyper path timed as a BURST of back-to-back calls between HIP events (the
per-launch table of profile_layers.py cannot tell what a node costs when its launches overlap on two streams).
python tools/encoder_segments.py [--batches 18x512x768,6x768x512] [--reps 6] [--autotune]"""
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
ap.add_argument("--batches", default="18x512x768,6x768x512")
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--config", default="two_layer_syn")
ap.add_argument("--autotune", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS[args.config]())
ops.side_streams(3, dev)


def burst(fn, reps, rounds):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


def node_flops(fn):
    ops.PROFILE = []
    fn()
    torch.cuda.synchronize()
    fl = sum(e["flops"] for e in ops.PROFILE)
    nl = len(ops.PROFILE)
    ops.PROFILE = None
    return fl, nl


def label(node):
    n = type(node).__name__
    name = getattr(node, "name", "")
    if n == "Conv":
        return f"Conv {name} k{node.k} s{node.s} ->{node.cout}"
    return f"{n} {name}"


for spec in args.batches.split(","):
    n, h, w = (int(v) for v in spec.split("x"))
    x = (torch.rand((n, h, w, 3), device=dev) - 0.5).contiguous()
    if args.autotune:
        with ops.autotune():
            model.encode(x, check=False)
    model.encode(x, check=False)
    torch.cuda.synchronize()
    print(f"== batch {n} x {h} x {w}")
    tot_ms = tot_fl = 0.0
    cur = x
    graph = model._analysis._graph
    for node in graph.layers:
        inp = cur
        fl, nl = node_flops(lambda: node(inp))
        ms = burst(lambda: node(inp), args.reps, args.rounds)
        cur = node(inp)
        tot_ms += ms; tot_fl += fl
        print(f"  {label(node):38s} in {tuple(inp.shape)!s:22s} {nl:3d} launches {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF {fl / 1e9:9.2f} GF")
    y = cur
    for name, tr, inp in (("hyper_analysis", model._hyper_analysis, y),):
        fl, nl = node_flops(lambda: tr(inp))
        ms = burst(lambda: tr(inp), args.reps, args.rounds)
        tot_ms += ms; tot_fl += fl
        print(f"  {name:38s} in {tuple(inp.shape)!s:22s} {nl:3d} launches {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF {fl / 1e9:9.2f} GF")
        z = tr(inp)
    z_hat, _ = model._get_prior()(z)
    fl, nl = node_flops(lambda: model._hyper_synthesis(z_hat))
    ms = burst(lambda: model._hyper_synthesis(z_hat), args.reps, args.rounds)
    tot_ms += ms; tot_fl += fl
    print(f"  {'hyper_synthesis':38s} in {tuple(z_hat.shape)!s:22s} {nl:3d} launches {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF {fl / 1e9:9.2f} GF")
    hyper = model._hyper_synthesis(z_hat)
    ms = burst(lambda: model._get_prior()(z), args.reps, args.rounds)
    print(f"  {'entropy z (factorized)':38s} {ms:8.4f} ms")
    tot_ms += ms
    ms = burst(lambda: ops.entropy_scale_normal(y, hyper, want_symbols=True), args.reps, args.rounds)
    print(f"  {'entropy y (scale normal)':38s} {ms:8.4f} ms")
    tot_ms += ms
    whole = burst(lambda: model.encode(x, check=False), 3, args.rounds)
    print(f"  sum of nodes {tot_ms:.3f} ms ({tot_fl / tot_ms / 1e9:.1f} TF); model.encode {whole:.3f} ms = {tot_fl / whole / 1e9:.1f} TFLOP/s "
          f"= {tot_fl / whole / 1e9 / 157.3:.3f} of peak")

#!/usr/bin/env python3
"""Time sntc_two_layer_out_adjoint against the gather-GEMM adjoint plan at the Tecnick SGA shape."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd import ops
dev = torch.device("cuda:0")
g = torch.randn((5, 1216, 1216, 3), device=dev)
w2 = torch.randn((5, 5, 3, 12), device=dev) * 0.1
plan = ops.ConvPlan("conv", w2, None, 2)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print(f"stream kernel {timed(lambda: ops.two_layer_out_adjoint(g, w2, 12)):.3f} ms, gather-GEMM {timed(lambda: plan(g)):.3f} ms, "
      f"traffic {2 * g.numel() * 4 / 1e6:.0f} MB")

#!/usr/bin/env python3
"""Cycles per phase of the two ResidualBlock kernels -- needs a DIAG build (make -C shallow-ntc_amd/csrc clean && make DIAG=1):
the kernel then overwrites the first floats of its output with s_memtime sums per workgroup (csrc/rb_fused_bf3.hip)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
c = 192
mk = lambda scale, *shape: torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).to(dev)
args = [mk(0.08, 1, 1, c, c // 2), mk(1, c // 2), mk(0.05, 3, 3, c // 2, c // 2), mk(1, c // 2), mk(0.1, 1, 1, c // 2, c), mk(1, c)]
x = mk(1.0, 18, 256, 384, c)
for precision, bound in (("fp32", (43.0, 165.9, 36.9)), ("bf16x3", (16.1, 62.2, 13.8))):
    plan = ops.ResBlockPlan(*args, precision=precision)
    for _ in range(3):
        y = plan(x)
    torch.cuda.synchronize()
    t = y.reshape(-1)[:1024].cpu().numpy().reshape(256, 4)
    m = (t[:, :3] / t[:, 3:4]).mean(0)
    print("%s: shader cycles per tile: head %.0f (MFMA-bound %.1f k)  3x3 %.0f (%.1f k)  tail %.0f (%.1f k)  total %.0f; tiles per workgroup %.1f"
          % (precision, m[0], bound[0], m[1], bound[1], m[2], bound[2], m.sum(), t[:, 3].mean()))

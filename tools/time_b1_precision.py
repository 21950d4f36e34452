#!/usr/bin/env python3
"""One image at a time (the reference's evaluate() pattern, mshyper/models.py:425-433) and small batches, fp32 against bf16x3:
where does the split-precision mode pay?  python tools/time_b1_precision.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

dev = torch.device("cuda:0")
cfg = configs.two_layer_syn()
models = {p: Model(device=dev, quality_metrics=False, precision=p, **cfg) for p in ("fp32", "bf16x3")}
w = models["fp32"].get_weights()
models["bf16x3"].set_weights(w)
for n in (1, 2, 4, 8, 18):
    x = (torch.rand((n, 512, 768, 3), device=dev) - 0.5).contiguous()
    row = []
    for p, m in models.items():
        z_hat, sym, _, _ = m.encode(x)
        for fn, label in ((lambda: m.decode(z_hat, sym, (512, 768), check=False), "decode"), (lambda: m.encode(x, check=False), "encode")):
            for _ in range(3):
                fn()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            row.append(f"{p} {label} {np.median(ts):7.3f} ms")
    print(f"n={n:2d}: " + " | ".join(row))

#!/usr/bin/env python3
"""The reference's evaluation flow (mshyper/models.py:425-433, eval_lib: one image at a time): wall time for a Kodak-shaped
set, strictly serial (lookahead 1) and with the default look-ahead grouping, PSNR only and with MS-SSIM."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
shapes = [(768, 512) if i in (3, 8, 9, 16, 17, 18) else (512, 768) for i in range(24)]
images = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(1, h, w, seed=i))).to(dev) for i, (h, w) in enumerate(shapes)]
px = sum(h * w for h, w in shapes)
for quality in (False, True):
    model._quality_metrics = quality
    for look in (1, 3):
        list(model.evaluate(images, lookahead=look))
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rows = [m.scalars_float for m in model.evaluate(images, lookahead=look)]
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))
        print(f"evaluate(lookahead={look}, msssim={quality}): {dt / len(rows) * 1e3:.3f} ms per image, {px / dt / 1e6:.1f} Mpixel/s", flush=True)

#!/usr/bin/env python3
"""The reference's evaluation flow (eval_lib: one image at a time, MS-SSIM on): wall time for a Kodak-shaped set."""
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
shapes = [(512, 768)] * 18 + [(768, 512)] * 6
images = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(1, h, w, seed=i))).to(dev) for i, (h, w) in enumerate(shapes)]
list(model.evaluate(images[:2]))
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = [m.scalars_float for m in model.evaluate(images)]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"evaluate(): {len(rows)} images one at a time in {dt * 1e3:.1f} ms ({dt / len(rows) * 1e3:.2f} ms per image, {sum(h * w for h, w in shapes) / dt / 1e6:.1f} Mpixel/s), keys {sorted(rows[0])}")

#!/usr/bin/env python3
"""How much of the strictly serial evaluate() (one image per pass, the reference's loop) is device time: run under
`rocprofv3 --kernel-trace --stats` and compare the summed kernel time with the wall clock this prints.
python tools/evaluate_busy.py [passes]"""
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
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
model._quality_metrics = False
images = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(1, 512, 768, seed=i))).to(dev) for i in range(8)]
list(model.evaluate(images, lookahead=1))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    rows = [m.scalars_float for m in model.evaluate(images, lookahead=1)]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("serial evaluate: %.3f ms per image wall over %d images (+ %d warm-up images whose kernels are in the trace too)"
      % (1e3 * dt / (passes * len(images)), passes * len(images), len(images)))

#!/usr/bin/env python3
"""Host (launch) time vs GPU time of one training forward + backward: is the step launch-bound?"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
from shallow_ntc_amd.train import Trainer
dev = torch.device("cuda:0")
cfg = configs.CONFIGS["two_layer_syn"]()
cfg["optimizer_config"] = dict(learning_rate=1e-4, global_clipnorm=1.0)
tr = Trainer(Model(device=dev, **cfg))
x = torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(8, 256, 256, seed=3))).to(dev)
with torch.cuda.device(dev):
    for _ in range(3):
        tr.loss_and_grads(x, 0.08)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        tr.loss_and_grads(x, 0.08)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
print(f"host launch time {1e3 * sorted(host)[5]:.2f} ms, until GPU done {1e3 * sorted(total)[5]:.2f} ms")

#!/usr/bin/env python3
"""Full-size two_layer_syn model, reference training shape: N optimizer steps on a fixed synthetic batch; prints the
loss curve (finite, decreasing) and the step rate.  python tools/train_sanity.py [--steps 200]"""
import argparse
import sys
import time
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

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--config", default="two_layer_syn")
ap.add_argument("--pool", type=int, default=1, help="number of distinct synthetic batches cycled through (1 = overfit one batch)")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = configs.CONFIGS[args.config]()
cfg["optimizer_config"] = dict(learning_rate=1e-4, global_clipnorm=1.0, warmup_steps=20)
cfg["scheduled_num_steps"] = args.steps
model = Model(device=dev, quality_metrics=False, **cfg)
pool = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(8, 256, 256, seed=3 + k))).to(dev) for k in range(args.pool)]
x = pool[0]
t0 = time.perf_counter()
for i in range(args.steps):
    m = model.train_step(pool[i % args.pool]).scalars_float
    if i % max(1, args.steps // 10) == 0 or i == args.steps - 1:
        print(f"step {i:4d}  rd_loss {m['rd_loss']:10.4f}  bpp {m['bpp']:.4f}  psnr {m['psnr']:.3f}  lr {m['scheduled_lr']:.2e}", flush=True)
torch.cuda.synchronize()
print(f"{args.steps / (time.perf_counter() - t0):.1f} steps/s")
model.trainer.sync_model()
rows = model.evaluate_batched(x)
print("eval after sync: bpp %.4f psnr %.3f" % (np.mean([r['bpp'] for r in rows]), np.mean([r['psnr'] for r in rows])))
blob = model.compress(x)
z_hat, sym, bits_z, bits_y = model.encode(x)
assert torch.equal(model.decompress(blob), model.decode(z_hat, sym, (256, 256)))
est = float(bits_z.sum() + bits_y.sum())
print(f"bitstream: {8 * len(blob)} bits on the wire vs {est:.0f} estimated ({8 * len(blob) / est - 1:+.2%}); "
      f"{8 * len(blob) / (x.shape[0] * 256 * 256):.4f} bpp real")

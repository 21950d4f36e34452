#!/usr/bin/env python3
"""R-D curve of a config over a lambda sweep on a Kodak-shaped set, sharded over the ranks of a node
(BASELINE.json configs[3]: mshyper/configs/jpegl.py, lambda in {0.001 ... 0.08}, batch sharded over 8 x MI355X).

    python tools/rd_sweep.py [--config jpegl] [--workdirs DIR ...] [--data-glob '/data/kodak/*.png'] [--out curve.json]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/rd_sweep.py ...

The work units are (lambda, image) pairs; unit u goes to rank u mod N, every rank holds one model per lambda it meets,
and ONE all-gather of the per-unit (bpp, psnr, mse, rd_loss) rows closes the run (no data-path collective).  Weights:
the latest checkpoint of a workdir whose run name carries that rd_lambda (--workdirs), otherwise framework-default
initial values (the curve is then a plumbing check, not a trained R-D curve).  Images: PNG files (--data-glob) or the
seeded synthetic Kodak-shaped set."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import distributed as D
from shallow_ntc_amd.common import data_lib, eval_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

LAMBDAS = [0.001, 0.0025, 0.005, 0.01, 0.02, 0.04, 0.08]          # mshyper/configs/jpegl.py get_hyper()

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="jpegl")
ap.add_argument("--lambdas", type=float, nargs="*", default=LAMBDAS)
ap.add_argument("--workdirs", nargs="*", default=[])
ap.add_argument("--data-glob", default=None)
ap.add_argument("--images", type=int, default=24)
ap.add_argument("--out", default=None)
args = ap.parse_args()

rank, local_rank, world = D.init()
dev = torch.device("cuda", 0 if __import__("os").environ.get("SNTC_SHARE_GPU") else local_rank)
torch.cuda.set_device(dev)
if args.data_glob:
    images = [b[0] for b in data_lib.get_dataset(args.data_glob, "val", 1, None)]
else:
    shapes = ([(512, 768)] * 18 + [(768, 512)] * 6)[:args.images]
    images = [data_lib.normalize_image(data_lib.synthetic_images(1, h, w, seed=100 + i))[0] for i, (h, w) in enumerate(shapes)]
units = [(li, ii) for li in range(len(args.lambdas)) for ii in range(len(images))]
by_lambda = {}
for d in args.workdirs:                                               # run names carry rd_lambda=<value>
    lam = eval_lib.parse_runname(Path(d).name, parse_numbers=True).get("rd_lambda")
    if lam is not None:
        by_lambda[float(lam)] = d
models = {}


def evaluate_unit(u):
    li, ii = units[u]
    lam = args.lambdas[li]
    if li not in models:
        if lam in by_lambda:
            models[li] = eval_lib.load_latest_ckpt(by_lambda[lam], device=dev, update_model_config=dict(quality_metrics=False))
        else:
            models[li] = Model(device=dev, quality_metrics=False, **configs.CONFIGS[args.config](rd_lambda=lam))
    m = models[li].validation_step(images[ii][None]).scalars_float
    return [m["bpp"], m["psnr"], m["mse"], m["rd_loss"]]


table = D.run_units(len(units), evaluate_unit, device=dev, width=4)
if rank == 0:
    curve = []
    for li, lam in enumerate(args.lambdas):
        t = table[li * len(images):(li + 1) * len(images)]
        curve.append(dict(rd_lambda=lam, bpp=float(t[:, 0].mean()), psnr=float(t[:, 1].mean()), mse=float(t[:, 2].mean()),
                          rd_loss=float(t[:, 3].mean()), images=len(images), trained=lam in by_lambda))
    out = json.dumps(dict(config=args.config, n_gpus=world, curve=curve), indent=1)
    if args.out:
        Path(args.out).write_text(out)
    print(out)
D.barrier()
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()

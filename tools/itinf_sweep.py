#!/usr/bin/env python3
"""Iterative inference (SGA) over a Tecnick-shaped set, sharded over the ranks of a node (BASELINE.json configs[4]:
mshyper/configs/two_layer_syn2.py + itinf.py on Tecnick-100, 8 x MI355X).

    python tools/itinf_sweep.py [--images 100] [--batch 5] [--hw 1200 1200] [--steps 3000] [--workdir DIR] [--out rows.json]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/itinf_sweep.py ...

The reference optimises the latents of every data batch independently (common/itinf_lib.py:187-207; the paper ran
Tecnick in batches of 5, results/readme.md:8): a batch is one work unit.  Unit u goes to rank u mod N
(distributed.run_units), the rank runs common/itinf_lib.itinf_on_data_batch on it -- num_steps SGA steps with the
hard-rounded evaluation every eval_every_steps -- and ONE all-gather of the final rows (bpp, psnr, mse, rd_loss, and
the same four before the optimisation) closes the run.  Weights: the latest checkpoint of --workdir, else
framework-default initial values (then a plumbing check, not a trained result)."""
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
from shallow_ntc_amd.common import data_lib, eval_lib, itinf_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=100)
ap.add_argument("--batch", type=int, default=5)
ap.add_argument("--hw", type=int, nargs=2, default=[1200, 1200])
ap.add_argument("--steps", type=int, default=3000)                 # mshyper/configs/itinf.py:24
ap.add_argument("--eval-every", type=int, default=200)            # :27
ap.add_argument("--rd-lambda", type=float, default=0.005)
ap.add_argument("--hidden", type=int, default=24)                 # two_layer_syn2.py:9-12 (FLOP-matched hidden width)
ap.add_argument("--workdir", default=None)
ap.add_argument("--operating-point", action="store_true",
                help="no checkpoint: rescale the initial weights so that the codec works near 1 bpp (tools/operating_point.py)")
ap.add_argument("--out", default=None)
args = ap.parse_args()

rank, local_rank, world = D.init()
dev = torch.device("cuda", 0 if __import__("os").environ.get("SNTC_SHARE_GPU") else local_rank)
torch.cuda.set_device(dev)
itinf = configs.itinf()
update = dict(latent_config=itinf["latent_config"], optimizer_config=itinf["optimizer_config"], scheduled_num_steps=args.steps,
              quality_metrics=False)
if args.workdir:
    model = eval_lib.load_latest_ckpt(args.workdir, device=dev, update_model_config=update)      # itinf_lib.py:175-176
else:
    cfg = configs.two_layer_syn2(rd_lambda=args.rd_lambda, hidden_channels=args.hidden)
    cfg.update(update)
    model = Model(device=dev, **cfg)
    if args.operating_point:
        sys.path.insert(0, str(ROOT / "tools"))
        import operating_point
        operating_point.apply(model)
h, w = args.hw
num_units = -(-args.images // args.batch)


def optimise_unit(u):
    ids = range(u * args.batch, min(args.images, (u + 1) * args.batch))
    x = np.concatenate([data_lib.normalize_image(data_lib.synthetic_images(1, h, w, seed=1000 + i)) for i in ids])
    before = model.validation_step(x).scalars_float
    _, val_rows, _ = itinf_lib.itinf_on_data_batch(dict(num_steps=args.steps, log_metrics_every_steps=max(1, args.steps // 10),
                                                        eval_every_steps=args.eval_every), None, None, model, x)
    after = val_rows[-1]
    # rd_loss = bpp + lambda * mse with THE run's lambda on both sides (validation_step outside itinf mode applies the
    # training-time lambda warm-up, mshyper/models.py:168-184, which would make "before" depend on what ran earlier)
    lam = model._rd_lambda
    return [after["bpp"], after["psnr"], after["mse"], after["bpp"] + lam * after["mse"], before["bpp"], before["psnr"], before["mse"],
            before["bpp"] + lam * before["mse"], len(ids)]


table = D.run_units(num_units, optimise_unit, device=dev, width=9)
if rank == 0:
    wts = table[:, 8] / table[:, 8].sum()
    avg = lambda c: float((table[:, c] * wts).sum())
    out = json.dumps(dict(config="two_layer_syn2 + itinf (SGA)", n_gpus=world, images=args.images, batch=args.batch, steps=args.steps,
                          sga=dict(bpp=avg(0), psnr=avg(1), mse=avg(2), rd_loss=avg(3)),
                          no_sga=dict(bpp=avg(4), psnr=avg(5), mse=avg(6), rd_loss=avg(7)),
                          units=[dict(unit=u, bpp=table[u, 0], psnr=table[u, 1], rd_loss=table[u, 3], rd_loss_before=table[u, 7],
                                      images=int(table[u, 8])) for u in range(num_units)]), indent=1)
    if args.out:
        Path(args.out).write_text(out)
    print(out)
D.barrier()
D.shutdown()

#!/usr/bin/env python3
"""Two-rank data-parallel training step == one full-batch step (run OUTSIDE pytest, launched before anything touches
the GPU):

    SNTC_SHARE_GPU=1 SNTC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29544 tools/ddp_check.py        # two ranks sharing one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/ddp_check.py   # RCCL
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from shallow_ntc_amd import distributed as D
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper.models import Model
from shallow_ntc_amd.train import Trainer

rank, local_rank, world = D.init()
dev = torch.device("cuda", 0 if os.environ.get("SNTC_SHARE_GPU") else local_rank)
torch.cuda.set_device(dev)
cfg = dict(analysis=dict(cls="ElicAnalysis", channels=(32, 32, 32, 32)),
           synthesis=dict(cls="TwoLayerResSynthesis", channels=(12, 3), strides=(8, 2), kernel_sizes=(13, 5), activation_type="igdn"))


def make():
    return Model(device=dev, rd_lambda=0.02, transform_config=cfg, scheduled_num_steps=1000,
                 optimizer_config=dict(learning_rate=1e-3, global_clipnorm=1.0, warmup_steps=0), quality_metrics=False)


n, h, w = 2 * world, 64, 64
x = data_lib.normalize_image(data_lib.synthetic_images(n, h, w, seed=21))
rng = np.random.default_rng(2)
nz = rng.uniform(-0.5, 0.5, size=(n, 1, 1, 32)).astype(np.float32)
ny = rng.uniform(-0.5, 0.5, size=(n, 4, 4, 32)).astype(np.float32)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sl = slice(2 * rank, 2 * rank + 2)
tr = Trainer(make(), seed=3)
tr.train_step(t(x[sl]), t(nz[sl]), t(ny[sl]))                  # sharded step with the bucketed all-reduce
got = tr.store.param.clone()
if rank == 0:
    import torch.distributed as dist
    was = dist.is_initialized()
ref = Trainer(make(), seed=3)
# the reference step must not all-reduce: compute the full-batch gradient locally and apply it by hand
out = ref.loss_and_grads(t(x), 0.02, t(nz), t(ny))
from shallow_ntc_amd import ops
import math
norm = math.sqrt(float(ops.sumsq(ref.store.grad).item()))
ops.adam_step(ref.store.param, ref.store.grad, ref.store.m, ref.store.v, 1e-3, 1, grad_scale=min(1.0, 1.0 / norm))
err = float((ref.store.param - got).abs().max())
print(f"rank {rank}: max |param(sharded step) - param(full-batch step)| = {err:.3e}", flush=True)
assert err < 5e-6
D.barrier()
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()

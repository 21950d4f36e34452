"""Evaluation harness (reference common/eval_lib.py:11-105, eval.py): load the latest checkpoint of a training
workdir into the MI355X Model and write the per-image results JSON with the reference's schema."""
from __future__ import annotations

import json
import re
from collections import OrderedDict
from pathlib import Path

from . import tf_checkpoint

TRAIN_COLLECTION = "train"            # common/train_lib.py:79-81
CHECKPOINTS_DIR_NAME = "checkpoints"


def parse_runname(s, parse_numbers=False):
    """'dir-lamb=2-arch=2_4_8/tau=1.0' -> OrderedDict(lamb='2', arch='2_4_8', tau='1.0'); with parse_numbers the
    values become ints / floats / int tuples (behaviour of reference common/utils.py:150-197)."""
    pattern = r"(\w+)=((\d+_)+\d+|(-?\d*\.?\d+(?:e[+-]?\d+)?)+|\w+)"
    res = OrderedDict()
    for m in re.finditer(pattern, s):
        key, val = m.group(1), m.group(2)
        if parse_numbers:
            if m.group(3) is not None:
                val = tuple(int(v) for v in val.split("_"))
            else:
                try:
                    f = float(val)
                    val = int(f) if f == int(f) else f
                except ValueError:
                    pass
        res[key] = val
    return res


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the 'checkpoint' state file names the newest prefix."""
    d = Path(checkpoint_dir)
    state = d / "checkpoint"
    if state.exists():
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', state.read_text())
        if m:
            p = Path(m.group(1))
            return p if p.is_absolute() else d / p
    cands = sorted(d.glob("ckpt-*.index"), key=lambda p: int(re.search(r"ckpt-(\d+)", p.name).group(1)))
    if not cands:
        raise FileNotFoundError(f"no checkpoint under {d}")
    return cands[-1].with_suffix("")


def load_latest_ckpt(workdir, model_cls=None, load_model_config=True, update_model_config=None, device=None):
    """reference eval_lib.py:11-53.  ``model_cls`` defaults to the mean-scale hyperprior Model (the reference
    re-imports the workdir's saved models.py; here the MI355X Model takes the same ``model_config``)."""
    from ..mshyper.models import Model
    workdir = Path(workdir)
    model_cls = model_cls or Model
    model_config = json.loads((workdir / "config.json").read_text())["model_config"] if load_model_config else {}
    model_config.update(update_model_config or {})
    model = model_cls(device=device, **model_config)
    prefix = latest_checkpoint(workdir / TRAIN_COLLECTION / CHECKPOINTS_DIR_NAME)
    model.set_weights(tf_checkpoint.load_reference_checkpoint(prefix, model_config["transform_config"]))
    model._step = int(re.search(r"ckpt-(\d+)", Path(prefix).name).group(1))
    return model


def results_rows(metrics_list, runname=""):
    """reference eval_lib.py:91-102: one flat dict per image = scalars + instance_id + run-name hparams."""
    hparams = parse_runname(runname, parse_numbers=True)
    rows = []
    for instance_id, m in enumerate(metrics_list):
        d = dict(m.scalars_float if hasattr(m, "scalars_float") else m)
        d["instance_id"] = instance_id
        d.update(hparams)
        rows.append(d)
    return rows


def eval_workdir(workdir, eval_data, results_dir, model_cls=None, skip_existing=True, device=None):
    """reference eval_lib.py:56-105: evaluate the latest checkpoint on ``eval_data`` (an iterable of
    [1,H,W,3] images or a [B,H,W,3] array) and dump ``{runname}-step={step}-xid={xid}.json``."""
    workdir = Path(workdir)
    results_dir = Path(results_dir)
    results_dir.mkdir(parents=True, exist_ok=True)
    runname = re.sub(r"^wid=\d+-", "", workdir.name)
    xid = workdir.parent.name
    model = load_latest_ckpt(workdir, model_cls=model_cls, device=device)
    path = results_dir / f"{runname}-step={int(model.global_step):3g}-xid={xid}.json"
    if path.exists() and skip_existing:
        return path
    rows = results_rows(list(model.evaluate(eval_data)), runname)
    path.write_text(json.dumps(rows, indent=2))
    return path

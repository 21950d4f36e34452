"""Metrics container (reference common/train_lib.py:22-76) and the train / eval loop (:87-258) around
``Model.train_step`` / ``validation_step``: same config keys, same cadence of logging, evaluation and checkpoints
(``train/checkpoints/ckpt-N`` in the reference's TensorBundle layout).  Logging goes to JSON-lines files
(``record.jsonl``, as the reference's custom writer also keeps, custom_writers.py:89-128); TensorBoard summaries and
the tf.data input pipeline are not reproduced -- ``train_dataset`` is any iterable of NHWC float batches."""
import json
import logging
import re
import time
from pathlib import Path
from typing import Any, Mapping, NamedTuple

TRAIN_COLLECTION = "train"            # :79-81
VAL_COLLECTION = "val"
CHECKPOINTS_DIR_NAME = "checkpoints"


log = logging.getLogger(__name__)


class Metrics(NamedTuple):
    scalars: Mapping[str, Any]
    images: Mapping[str, Any]

    @classmethod
    def make(cls):
        return Metrics(scalars={}, images={})

    def record_scalar(self, key, value):
        self.scalars[key] = value

    def record_scalars(self, scalars):
        for key, value in scalars.items():
            self.scalars[key] = value

    def record_image(self, key, value):
        self.images[key] = value

    @property
    def scalars_numpy(self):
        return {k: float(v) for k, v in self.scalars.items()}

    @property
    def scalars_float(self):
        return {k: float(v) for k, v in self.scalars.items()}

    @classmethod
    def merge_metrics(cls, metrics_list):
        """Mean of scalars across batches (train_lib.py:58-76); images are dropped."""
        keys = metrics_list[0].scalars.keys()
        merged = {k: sum(float(m.scalars[k]) for m in metrics_list) / len(metrics_list) for k in keys}
        return Metrics(scalars=merged, images={})


class JsonlWriter:
    """write_scalars(step, {name: value}) -> one JSON object per line in <dir>/record.jsonl."""

    def __init__(self, directory):
        self.dir = Path(directory)
        self.dir.mkdir(parents=True, exist_ok=True)
        self._f = open(self.dir / "record.jsonl", "a")

    def write_scalars(self, step, scalars):
        self._f.write(json.dumps(dict(step=int(step), time=time.time(), **{k: float(v) for k, v in scalars.items()})) + "\n")
        self._f.flush()

    def write_hparams(self, hparams):
        (self.dir / "hparams.json").write_text(json.dumps(hparams, indent=1, default=str))

    def write_images(self, step, images):
        pass

    def close(self):
        self._f.close()


def _cfg(config, key, default=None):
    return config.get(key, default) if hasattr(config, "get") else getattr(config, key, default)


def simple_train_eval_loop(train_eval_config, workdir, model, train_dataset, val_data):
    """reference :87-258.  ``train_eval_config``: num_steps, log_metrics_every_steps, checkpoint_every_steps,
    eval_every_steps[, warm_start: a checkpoint prefix / workdir].  Returns the list of logged train rows."""
    from . import eval_lib, tf_checkpoint
    config = train_eval_config
    num_steps = int(_cfg(config, "num_steps"))
    log_every = int(_cfg(config, "log_metrics_every_steps", 100))
    ckpt_every = int(_cfg(config, "checkpoint_every_steps", num_steps))
    eval_every = int(_cfg(config, "eval_every_steps", 0))
    workdir = Path(workdir)
    train_writer, val_writer = JsonlWriter(workdir / TRAIN_COLLECTION), JsonlWriter(workdir / VAL_COLLECTION)
    train_writer.write_hparams(dict(config) if hasattr(config, "keys") else {})
    warm = _cfg(config, "warm_start")
    if warm:                                                   # :131-187 (prefix, or a dir holding checkpoints / train/checkpoints)
        warm = Path(warm)
        prefix = warm if not warm.is_dir() else (eval_lib.latest_checkpoint(warm) if list(warm.glob("ckpt-*.index"))
                                                  else eval_lib.latest_checkpoint(warm / TRAIN_COLLECTION / CHECKPOINTS_DIR_NAME))
        model.set_weights(tf_checkpoint.load_reference_checkpoint(prefix, model._transform_config))
        log.info("warm start: variables restored from %s (step stays %d; the reference also restores its global_step and Adam "
                 "slots from a warm checkpoint: continue in the same workdir to keep them)", prefix, int(model._step))
    # restore_or_initialize (:190): a workdir that already holds a checkpoint continues from it -- variables, step (hence
    # the lr / lambda schedules and Adam's bias correction) and, when this build wrote it, the Adam moments
    own = workdir / TRAIN_COLLECTION / CHECKPOINTS_DIR_NAME
    if (own / "checkpoint").exists() or list(own.glob("ckpt-*.index")):
        prefix = eval_lib.latest_checkpoint(own)
        model.set_weights(tf_checkpoint.load_reference_checkpoint(prefix, model._transform_config))
        model._step = int(re.search(r"ckpt-(\d+)", Path(prefix).name).group(1))
        from ..train import Trainer
        model.trainer = Trainer(model, seed=model._seed)
        if warm:
            log.warning("the workdir's own checkpoint %s overrides the warm start", prefix)
        if model.trainer.restore_optimizer(prefix):
            log.info("resumed from %s: variables, step %d, Adam moments", prefix, int(model._step))
        else:
            log.warning("resumed from %s at step %d WITHOUT optimizer state (no %s.optimizer.npz: a checkpoint of the reference, or "
                        "variables only): Adam continues with fresh moments", prefix, int(model._step), Path(prefix).name)
    rows = []

    def evaluate_fn(step):                                     # :214-229
        metrics = Metrics.merge_metrics([model.validation_step(batch) for batch in val_data])
        val_writer.write_scalars(step, metrics.scalars_float)

    it = iter(train_dataset)
    step = int(model.global_step)
    while step < num_steps:                                    # :233-250
        metrics = model.train_step(next(it))
        if step % log_every == 0:
            train_writer.write_scalars(step, metrics.scalars_float)
            rows.append(dict(step=step, **metrics.scalars_float))
        step += 1
        evaluating = eval_every > 0 and step % eval_every == 0 and step < num_steps
        saving = step % ckpt_every == 0 or step == num_steps
        if evaluating or saving:
            model.trainer.sync_model()
        if evaluating:
            evaluate_fn(step)
        if saving:
            model.trainer.save_checkpoint(workdir, max_to_keep=int(_cfg(config, "max_ckpts_to_keep", 1)))
    if eval_every > 0:                                         # :254-258 final evaluation
        model.trainer.sync_model()
        evaluate_fn(step)
    train_writer.close()
    val_writer.close()
    return rows

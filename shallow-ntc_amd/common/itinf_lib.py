"""Iterative-inference (SGA) loop on one data batch (reference common/itinf_lib.py:26-93): same cadence of logging and
hard-rounded evaluations, returns (train rows, validation rows, final latent variables as arrays)."""
from __future__ import annotations


def _cfg(config, key, default=None):
    return config.get(key, default) if hasattr(config, "get") else getattr(config, key, default)


def itinf_on_data_batch(train_eval_config, train_writer, val_writer, model, data_batch):
    """``train_eval_config``: num_steps, log_metrics_every_steps, eval_every_steps (mshyper/configs/itinf.py:21-33).
    Writers need ``write_scalars(step, dict)`` (e.g. train_lib.JsonlWriter) or may be None."""
    num_steps = int(_cfg(train_eval_config, "num_steps"))
    log_every = int(_cfg(train_eval_config, "log_metrics_every_steps", 100))
    eval_every = int(_cfg(train_eval_config, "eval_every_steps", 0))
    train_metrics, val_metrics = [], []

    def evaluate_fn(step):                                     # :53-60
        metrics = model.itinf_validation_step(data_batch, training=False)
        if val_writer is not None:
            val_writer.write_scalars(step, metrics.scalars_float)
        val_metrics.append({"step": step, **metrics.scalars_float})

    model.initialize_itinf(data_batch)                         # :62
    step = 0
    while step < num_steps:                                    # :67-82
        # the step's scalars stay on the device unless this step is logged (:67-75: the tf.function step returns tensors and
        # only the logged ones are converted): no host synchronisation inside the 3000-step loop
        metrics = model.itinf_train_step(data_batch, fetch=step % log_every == 0)
        if step % log_every == 0:
            if train_writer is not None:
                train_writer.write_scalars(step, metrics.scalars_float)
            train_metrics.append({"step": step, **metrics.scalars_float})
        step += 1
        if eval_every > 0 and step % eval_every == 0 and step < num_steps:
            evaluate_fn(step)
    # the reference's step runs check_numerics every step (mshyper/models.py:308-309,356); here the unlogged steps leave their
    # scalars on the device, so the last step's loss and the launch status are checked once more before anything is returned
    model.itinf_last_metrics()
    if eval_every > 0:                                         # :86-90
        evaluate_fn(step)
    lat = model.latent_rvs                                     # :92 the optimised variables as arrays
    itinf_vars = {"z_loc": lat.uq[0].loc.cpu().numpy(), "y_loc": lat.uq[1].loc.cpu().numpy()}
    return train_metrics, val_metrics, itinf_vars

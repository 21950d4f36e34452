"""Latent random variables (reference common/latent_rvs_lib.py:59-166): a uniformly-quantised latent with a location
parameter, and the (z, y) collection.  The arithmetic of ``sample`` / ``quantize`` is one element-wise HIP kernel
(``sntc_uq_sample``); the fused entropy kernels (ops.DeepFactorizedPrior / ops.entropy_scale_normal) implement the same
``round(loc - offset) + offset`` together with the rate."""
from __future__ import annotations

from typing import Any, Mapping, NamedTuple, Optional, Sequence

from .. import ops


class UQLatentRV:
    """A continuous latent expected to be rounded (latent_rvs_lib.py:59-116).  ``loc`` is a CUDA NHWC float32 tensor."""

    def __init__(self, loc):
        self._params = dict(loc=loc)

    @property
    def loc(self):
        return self._params["loc"]

    @property
    def params(self):
        return self._params

    @property
    def shape(self):
        return self.loc.shape

    def get_trainable_copy(self):
        """Copy whose ``loc`` is a fresh buffer that iterative inference may update in place (:44-55)."""
        return UQLatentRV(self.loc.clone())

    @property
    def trainable_variables(self):
        return [self.loc]

    def quantize(self, offset=None):
        """tfc.round_st(loc, offset) (:77-78): forward value round(loc - offset) + offset."""
        return ops.uq_sample(self.loc, offset, "round")

    def sample(self, training: bool, method: Optional[str] = None, offset=None, noise=None, seed=0, step=0, **kwargs):
        """:80-116.  ``training=False``: round(loc - offset) + offset whatever the method.  Otherwise 'unoise' (loc +
        U(-.5, .5)), 'sga' (kwargs['tau']; stochastic Gumbel annealing around ``offset``) or 'soft_round' (kwargs['alpha']).
        ``noise`` (uniform values / Gumbel pairs [..., 2]) makes a draw reproducible; otherwise (seed, step) key the
        counter-based generator.  Unused keys of the reference's latent_config (tau_r, tau_ub, ...) are accepted."""
        if not training:
            return ops.uq_sample(self.loc, offset, "round")
        if method == "unoise":
            return ops.uq_sample(self.loc, None, "unoise", 0.0, noise, seed, step)
        if method == "sga":
            return ops.uq_sample(self.loc, offset, "sga", kwargs["tau"], noise, seed, step)
        if method == "soft_round":
            return ops.uq_sample(self.loc, offset, "soft_round", kwargs["alpha"])
        raise NotImplementedError(method)


class LatentRVSamples(NamedTuple):
    """latent_rvs_lib.py:123-127."""
    uq: Sequence = tuple()
    categorical: Sequence = tuple()


class LatentRVCollection(NamedTuple):
    """latent_rvs_lib.py:130-166."""
    uq: Sequence[UQLatentRV] = tuple()
    categorical: Sequence = tuple()

    def sample(self, training, latent_config: Mapping[str, Any] = {}) -> LatentRVSamples:
        """:137-155: every rv of a kind is sampled with that kind's config."""
        cfg = dict(latent_config.get("uq", {}))
        return LatentRVSamples(uq=[rv.sample(training, **cfg) for rv in self.uq], categorical=[])

    def get_trainable_copy(self):
        return LatentRVCollection(uq=tuple(rv.get_trainable_copy() for rv in self.uq), categorical=self.categorical)

    @property
    def trainable_variables(self):
        return [rv.loc for rv in self.uq]

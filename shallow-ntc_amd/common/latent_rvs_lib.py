"""Latent random variables (reference common/latent_rvs_lib.py:59-166), reduced to what the hot
path uses: a uniformly-quantised latent with a location parameter, and the (z, y) collection."""
from __future__ import annotations

from typing import NamedTuple, Sequence


class UQLatentRV:
    """A continuous latent expected to be rounded (latent_rvs_lib.py:59-116).  ``loc`` is a CUDA
    NHWC float32 tensor.  Quantisation itself happens inside the fused entropy kernels
    (ops.DeepFactorizedPrior / ops.entropy_scale_normal), which implement
    ``round(loc - offset) + offset`` (:95-102)."""

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


class LatentRVCollection(NamedTuple):
    """latent_rvs_lib.py:130-166."""
    uq: Sequence[UQLatentRV] = tuple()
    categorical: Sequence = tuple()

    def get_trainable_copy(self):
        return LatentRVCollection(uq=tuple(rv.get_trainable_copy() for rv in self.uq), categorical=self.categorical)

    @property
    def trainable_variables(self):
        return [rv.loc for rv in self.uq]

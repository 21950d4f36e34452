"""Factorized-prior model (reference factorized/models.py:39-183): analysis -> round + deep-factorized
bits -> synthesis.  Same API as mshyper.models.Model; only the three overridden methods differ."""
from __future__ import annotations

import torch

from .. import ops
from ..common import image_utils
from ..common.latent_rvs_lib import LatentRVCollection, UQLatentRV
from ..mshyper import models as _ms
from ..mshyper.models import EMPTY_DICT, deep_factorized_init

CODING_RANK = 3
DOWNSAMPLE_FACTOR = 16   # reference :30, hard-coded
MIN_IMAGE_DIM = 32       # reference :31


class Model(_ms.Model):
    factorized = True

    def _init_transforms(self, transform_config=EMPTY_DICT):           # reference :51-68
        self._analysis = self._build_named(transform_config["analysis"], 3, 0)
        self._bottleneck_size = b = self._analysis.out_channels(3)
        self._synthesis = self._build_named(transform_config["synthesis"], b, 1)
        self._hyper_analysis = self._hyper_synthesis = None
        self._hyper_bottleneck_size = None
        self._prior_weights = deep_factorized_init(b, self._prior_num_filters, seed=self._seed + 4)
        self.downsample_factor = DOWNSAMPLE_FACTOR
        for t in self._transforms().values():
            t.build(device=self.device)

    def _transforms(self):
        from collections import OrderedDict
        return OrderedDict(analysis=self._analysis, synthesis=self._synthesis)

    def _prior_channels(self):
        return self._bottleneck_size

    def infer_latent_rvs(self, x):                                      # reference :70-87
        x = self._as_device_images(x)
        with torch.cuda.device(self.device):
            y = self._analysis(image_utils.pad_images(x, DOWNSAMPLE_FACTOR))
        return LatentRVCollection(uq=(UQLatentRV(y),))

    def _rate_and_reconstruction(self, latent_rvs, want_symbols=False):  # reference :101-124
        y = latent_rvs.uq[0].loc
        y_hat, bits = self._get_prior()(y)
        recon = self._synthesis(y_hat)
        return dict(z_hat=None, y_hat=y_hat, symbols=None, hyper=None, bits_z=torch.zeros_like(bits), bits_y=bits,
                    recon=recon)

    def _training_samples(self, latent_rvs, noise, seed):                # reference :101-118, training=True
        uq = self._latent_config["uq"].get("method", "unoise")
        step = self.global_step
        ny = None if noise is None else noise[-1]
        (y_rv,) = latent_rvs.uq
        prior = self._get_prior()
        if uq in ("unoise", "mixedq"):
            y_t = y_rv.sample(True, "unoise", noise=ny, seed=seed, step=2 * step + 1)
            bits, _ = ops.noisy_factorized(prior, y_t)
            return None, bits, (y_rv.quantize() if uq == "mixedq" else y_t)
        cfg = dict(self.latent_config["uq"])
        if uq == "sga":
            y_t, _, _, bits = ops.sga_factorized_fwd(prior, y_rv.loc, cfg["tau"], ny, seed, step)
            return None, bits, y_t
        y_t = y_rv.sample(True, offset=None, noise=ny, seed=seed, step=2 * step + 1, **cfg)
        bits, _ = ops.noisy_factorized(prior, y_t)
        return None, bits, y_t

    def encode(self, x, check=True):
        lat = self.infer_latent_rvs(self._as_device_images(x))
        y_hat, bits = self._get_prior()(lat.uq[0].loc)
        if check:
            with torch.cuda.device(self.device):
                ops.check_conv_status()
        return y_hat, None, None, bits

    def decode(self, y_hat, symbols, image_hw, reference=None, check=True):
        with torch.cuda.device(self.device):
            recon = self._synthesis(y_hat)
            if reference is None:
                out = ops.to_pixels(recon, image_hw[0], image_hw[1])
            else:
                sse, px = ops.pixels_sse(reference, recon, want_pixels=True)
                out = (px, sse)
            if check:
                ops.check_conv_status()
        return out

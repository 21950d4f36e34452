"""float64 NumPy restatement of the reference ``Model`` eval forward (and the SGA loss).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Follows reference mshyper/models.py:111-140 (transform construction), :212-232
(infer_latent_rvs), :234-359 (frame_loss_given_latent_rvs, eval branch and the explicit-sampling
SGA branch) and factorized/models.py:51-183.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from . import ops_np as ops
from . import transforms_np as T

DUMMY_IMG_DIM = 64          # mshyper/models.py:37
FACTORIZED_DOWNSAMPLE = 16  # factorized/models.py:30


def deep_factorized_shapes(channels, num_filters=(3, 3)):
    """tfc.DeepFactorized variables: matrices [C,f_{k+1},f_k], biases [C,f_{k+1}], factors [C,f_{k+1}]."""
    filters = (1,) + tuple(num_filters) + (1,)
    d = OrderedDict()
    for k in range(len(filters) - 1):
        d[f"prior/matrix_{k}"] = (channels, filters[k + 1], filters[k])
        d[f"prior/bias_{k}"] = (channels, filters[k + 1])
        if k < len(filters) - 2:
            d[f"prior/factor_{k}"] = (channels, filters[k + 1])
    return d


def init_deep_factorized(channels, rng, num_filters=(3, 3), init_scale=10.0):
    """TFC initialisers: matrix = log(expm1(1/scale/f_{k+1})), bias ~ U(-.5,.5), factor = 0."""
    filters = (1,) + tuple(num_filters) + (1,)
    scale = init_scale ** (1.0 / (len(num_filters) + 1))
    out = OrderedDict()
    for name, shp in deep_factorized_shapes(channels, num_filters).items():
        kind, k = name.split("/")[1].rsplit("_", 1)
        k = int(k)
        if kind == "matrix":
            v = np.full(shp, np.log(np.expm1(1.0 / scale / filters[k + 1])))
        elif kind == "bias":
            v = rng.uniform(-0.5, 0.5, size=shp)
        else:
            v = np.zeros(shp)
        out[name] = v.astype(np.float32)
    return out


def _prior_lists(params):
    ms, bs, fs = [], [], []
    k = 0
    while f"prior/matrix_{k}" in params:
        ms.append(params[f"prior/matrix_{k}"])
        bs.append(params[f"prior/bias_{k}"])
        if f"prior/factor_{k}" in params:
            fs.append(params[f"prior/factor_{k}"])
        k += 1
    return ms, bs, fs


class Model:
    """Mean-scale hyperprior (factorized=False) or factorized prior (factorized=True)."""

    def __init__(self, transform_config, rd_lambda=0.01, factorized=False, num_filters=(3, 3)):
        self.rd_lambda = rd_lambda
        self.factorized = factorized
        self.num_filters = tuple(num_filters)
        a = dict(transform_config["analysis"])
        self.analysis = T.build(a.pop("cls"), **a)
        # bottleneck size: run the shapes only (mshyper/models.py:117-119)
        self.bottleneck = self._out_channels(self.analysis)
        s = dict(transform_config["synthesis"])
        self.synthesis = T.build(s.pop("cls"), cin=self.bottleneck, **s)
        if factorized:
            self.hyper_analysis = self.hyper_synthesis = None
            self.downsample_factor = FACTORIZED_DOWNSAMPLE
            self.prior_channels = self.bottleneck
        else:
            ha = dict(transform_config.get("hyper_analysis",
                                           dict(cls="HyperAnalysis", bottleneck_size=self.bottleneck)))
            hs = dict(transform_config.get("hyper_synthesis",
                                           dict(cls="HyperSynthesis", bottleneck_size=self.bottleneck)))
            self.hyper_analysis = T.build(ha.pop("cls"), cin=self.bottleneck, **ha)
            self.prior_channels = self._out_channels(self.hyper_analysis)
            self.hyper_synthesis = T.build(hs.pop("cls"), cin=self.prior_channels, **hs)
            # downsample_factor = 64 / dummy hyper-latent dim (mshyper/models.py:137-140)
            dim = DUMMY_IMG_DIM
            for t in (self.analysis, self.hyper_analysis):
                dim = self._spatial(t, dim)
            self.downsample_factor = DUMMY_IMG_DIM // dim

    @staticmethod
    def _out_channels(t):
        return t.graph.shapes(t.cin)[1]

    @staticmethod
    def _spatial(t, dim):
        return t.graph.flops(t.cin, dim, dim)[2]

    # ---- parameters -------------------------------------------------------------------
    def param_shapes(self):
        d = OrderedDict()
        parts = [("analysis/", self.analysis), ("synthesis/", self.synthesis)]
        if not self.factorized:
            parts += [("hyper_analysis/", self.hyper_analysis), ("hyper_synthesis/", self.hyper_synthesis)]
        for pre, t in parts:
            for k, v in t.param_shapes().items():
                d[pre + k] = v
        d.update(deep_factorized_shapes(self.prior_channels, self.num_filters))
        return d

    def init_params(self, seed=4321):
        rng = np.random.default_rng(seed)
        shapes = self.param_shapes()
        p = T.init_params(OrderedDict((k, v) for k, v in shapes.items() if not k.startswith("prior/")), rng)
        p.update(init_deep_factorized(self.prior_channels, rng, self.num_filters))
        return p

    # ---- inference path (mshyper/models.py:212-232; factorized/models.py:70-87) ---------
    def _run(self, transform, params, prefix, x, be):
        """One transform, on the float64 NumPy ops (be None) or on a float64 backend module that offers
        ``as_params`` (oracle/train_ref: library convolutions, for full-size images); NHWC float64 in and out."""
        sub = T.sub_params(params, prefix)
        if be is None:
            return transform(sub, x)
        import torch
        with torch.no_grad():
            return be.to_nhwc(transform(be.as_params(sub), x, be=be))

    def infer_latents(self, params, x, be=None):
        xp = ops.pad_images(np.asarray(x, np.float64), self.downsample_factor)
        y = self._run(self.analysis, params, "analysis/", xp, be)
        if self.factorized:
            return (y,)
        z = self._run(self.hyper_analysis, params, "hyper_analysis/", y, be)
        return (z, y)

    # ---- generative path + losses, eval branch (mshyper/models.py:234-359) ---------------
    def frame_loss(self, params, x, latents, sga=None, be=None, force_symbols=None, force_z=None):
        """sga=None: training=False hard rounding.  sga=dict(tau, gumbel_z, gumbel_y): the explicit-
        sampling training branch used by itinf_train_step (models.py:260-268,285-291).
        force_symbols (tests only): integer symbols to use INSTEAD of round(y - mu), so that the rate and the
        reconstruction of another implementation's symbols can be checked separately from the (counted) positions
        where its float32 y - mu fell on the other side of a rounding boundary; ``tie_distance`` then reports
        | |frac(y - mu)| - 0.5 | of the oracle at every position.
        force_z (tests only): integer hyper-latents to use INSTEAD of round(z) -- a float32 implementation whose z lies within
        ~1e-6 of a rounding tie can land on the other side, which moves mu / sigma over that hyper-latent's receptive field; with
        z pinned the rest of the chain is compared like for like, and ``z_tie_distance`` reports | |frac(z)| - 0.5 | of the oracle."""
        x = np.asarray(x, np.float64)
        ms, bs, fs = _prior_lists(params)
        ln2 = math.log(2.0)
        out = {}
        if self.factorized:
            (y_loc,) = latents
            if sga is None:
                y_hat, bits = ops.batched_deep_factorized(y_loc, ms, bs, fs)
            else:
                y_hat = ops.sga_round(y_loc, sga["tau"], sga["gumbel_y"], offset=None)
                bits = ops.deep_factorized_logprob(y_hat, ms, bs, fs).sum(axis=(1, 2, 3)) / -ln2
            out["symbols_y"] = y_hat
            bits_z = None
            bits_y = bits
        else:
            z_loc, y_loc = latents
            if sga is None and force_z is not None:
                z_hat = np.asarray(force_z, np.float64)
                bits_z = ops.deep_factorized_logprob(z_hat, ms, bs, fs).sum(axis=(1, 2, 3)) / -ln2
                dz = np.asarray(z_loc, np.float64)
                out["z_tie_distance"] = np.abs(np.abs(dz - np.floor(dz) - 0.5))
            elif sga is None:
                z_hat, bits_z = ops.batched_deep_factorized(z_loc, ms, bs, fs)
            else:
                z_hat = ops.sga_round(z_loc, sga["tau"], sga["gumbel_z"], offset=0.0)
                bits_z = ops.deep_factorized_logprob(z_hat, ms, bs, fs).sum(axis=(1, 2, 3)) / -ln2
            h = self._run(self.hyper_synthesis, params, "hyper_synthesis/", z_hat, be)
            c = h.shape[-1] // 2
            mu, raw = h[..., :c], h[..., c:]
            indexes = np.exp(raw)                       # models.py:274-276 (sigma used as scale INDEX)
            if sga is None and force_symbols is not None:
                sym = np.asarray(force_symbols, np.float64)
                sigma = ops.scale_fn(np.clip(indexes, 0.0, ops.NUM_SCALES - 1.0))
                y_hat = sym + mu
                bits_y = ops.noisy_normal_logprob(sym, sigma).sum(axis=(1, 2, 3)) / -ln2
                d = np.asarray(y_loc, np.float64) - mu
                out["symbols_y"], out["tie_distance"] = sym, np.abs(np.abs(d - np.floor(d) - 0.5))
            elif sga is None:
                y_hat, bits_y, sym = ops.scale_indexed_normal(y_loc, mu, indexes)
                out["symbols_y"] = sym
            else:
                y_hat = ops.sga_round(y_loc, sga["tau"], sga["gumbel_y"], offset=mu)
                sigma = ops.scale_fn(np.clip(indexes, 0.0, ops.NUM_SCALES - 1.0))
                bits_y = ops.noisy_normal_logprob(y_hat - mu, sigma).sum(axis=(1, 2, 3)) / -ln2
            out.update(z_hat=z_hat, mu=mu, indexes=indexes)
        recon = self._run(self.synthesis, params, "synthesis/", y_hat, be)
        recon = ops.unpad_images(recon, x.shape)
        npix = float(x.shape[1] * x.shape[2])
        bpp = (0.0 if bits_z is None else bits_z.mean() / npix) + bits_y.mean() / npix
        training = sga is not None
        xp = ops.floats_to_pixels(x, training)
        rp = ops.floats_to_pixels(recon, training)
        mses, psnrs = ops.mse_psnr(xp, rp)
        mse, psnr = mses.mean(), psnrs.mean()
        out.update(y_hat=y_hat, recon=recon, recon_pixels=rp, bits_z=bits_z, bits_y=bits_y,
                   bpp=bpp, mse=mse, psnr=psnr, mses=mses, psnrs=psnrs,
                   rd_loss=bpp + self.rd_lambda * mse)
        return out

    def end_to_end(self, params, x, be=None):
        return self.frame_loss(params, x, self.infer_latents(params, x, be=be), be=be)

    def evaluate(self, params, images):
        """models.py:415-433: one image at a time."""
        for i in range(len(images)):
            yield self.end_to_end(params, images[i:i + 1])

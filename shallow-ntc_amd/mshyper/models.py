"""Mean-scale hyperprior model: the reference's ``Model`` API on the MI355X kernels.

Mirrors reference mshyper/models.py: constructor arguments (:46-51), ``infer_latent_rvs`` (:212-232),
``frame_loss_given_latent_rvs`` (:234-359), ``end_to_end_frame_loss`` (:361-373), ``validation_step``
(:385-387), ``evaluate`` (:415-433), ``downsample_factor`` (:137-140) and the ``Metrics.scalars`` keys
(:342-354: rd_loss, bpp, mse, psnr, scheduled_lr, sched_rd_lambda).  MS-SSIM / LPIPS (:321-340) and the
training step (:375-383) are out of scope.

Additionally exposes the codec regions SURVEY.md 8(d) measures: ``encode`` (x -> z_hat, symbols, bits)
and ``decode`` ((z_hat, symbols) -> uint8 pixels).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from .. import _capi as capi
from .. import ops
from ..common import data_lib, image_utils
from ..common.latent_rvs_lib import LatentRVCollection, UQLatentRV
from ..common.latent_rvs_utils import sga_schedule_at_step
from ..common.train_lib import Metrics
from ..common.transforms import class_builder as transform_builder

EMPTY_DICT = {}

# Fixed configs for the ScaleIndexedEntropyModel (reference :28-34); applied inside
# csrc/entropy.hip (kLogScaleMin, kScaleFactor).
NUM_SCALES = 64
SCALE_MIN = 0.11
SCALE_MAX = 256.0
SCALE_FACTOR = (math.log(SCALE_MAX) - math.log(SCALE_MIN)) / (NUM_SCALES - 1.0)
CODING_RANK = 3
DUMMY_IMG_DIM = 64
HIGHER_LAMBDA_UNTIL = 0.2
HIGHER_LAMBDA_FACTOR = 10.0


def deep_factorized_shapes(channels, num_filters=(3, 3)):
    filters = (1,) + tuple(num_filters) + (1,)
    d = OrderedDict()
    for k in range(len(filters) - 1):
        d[f"prior/matrix_{k}"] = (channels, filters[k + 1], filters[k])
        d[f"prior/bias_{k}"] = (channels, filters[k + 1])
        if k < len(filters) - 2:
            d[f"prior/factor_{k}"] = (channels, filters[k + 1])
    return d


def deep_factorized_init(channels, num_filters=(3, 3), init_scale=10.0, seed=4321):
    """tfc.DeepFactorized initial values: matrix = log(expm1(1/scale/f_{k+1})), bias ~ U(-.5,.5), factor = 0."""
    rng = np.random.default_rng(seed)
    filters = (1,) + tuple(num_filters) + (1,)
    scale = init_scale ** (1.0 / (len(num_filters) + 1))
    out = OrderedDict()
    for name, shp in deep_factorized_shapes(channels, num_filters).items():
        kind, k = name.split("/")[1].rsplit("_", 1)
        if kind == "matrix":
            v = np.full(shp, np.log(np.expm1(1.0 / scale / filters[int(k) + 1])))
        elif kind == "bias":
            v = rng.uniform(-0.5, 0.5, size=shp)
        else:
            v = np.zeros(shp)
        out[name] = v.astype(np.float32)
    return out


def compression_lr(optimizer_config, scheduled_num_steps, step):
    """Learning rate the reference's Adam uses for the update taken at optimizer iteration ``step`` (0-based):
    CompressionSchedule (common/schedule.py:155-176 via mshyper/models.py:95-108) = base * piecewise-constant
    [1, reduce_lr_factor] switching at int(reduce_lr_after * total) (boundary <= step picks the second value,
    schedule.py:46-48) * linear warm-up min(1, (step + 1) / warmup_steps) (schedule.py:121-123 -- the `+ 1` makes the
    very first update non-zero, and the warm-up factor also multiplies the dropped value)."""
    cfg = optimizer_config
    lr = cfg.get("learning_rate", 1e-4)
    after = cfg.get("reduce_lr_after", 0.8)
    factor = cfg.get("reduce_lr_factor", 0.1)
    warmup = cfg["warmup_steps"] if "warmup_steps" in cfg else int(cfg.get("warmup_until", 0.02) * scheduled_num_steps)
    value = lr * (factor if step >= int(after * scheduled_num_steps) else 1.0)
    if warmup > 0:
        value *= min(1.0, (step + 1) / warmup)
    return value


class Model:
    """Encapsulates the transforms + entropy models (reference :45-149)."""

    factorized = False

    def __init__(self, scheduled_num_steps=1500000, rd_lambda=0.01, offset_heuristic=True,
                 transform_config=EMPTY_DICT, optimizer_config=EMPTY_DICT,
                 latent_config=None, profile=False, device=None, prior_num_filters=(3, 3), seed=4321,
                 quality_metrics=True, precision="fp32"):
        """``precision``: "fp32" (default: exact fp32 MFMA everywhere, the reference's arithmetic) or "bf16x3": the
        convolutions that qualify (Cin % 16 == 0, at least one 256-row strip per image) run the split-precision contraction
        on pre-split operands (csrc/bf3_gemm.hip; ~fp32 accuracy, not bit-identical to it).  An encoder and a decoder must
        use the same precision: the bitstream header carries it."""
        capi.require_gpu()
        self._scheduled_num_steps = scheduled_num_steps
        self._rd_lambda = rd_lambda
        self._latent_config = dict(latent_config) if latent_config is not None else dict(uq=dict(method="unoise"))
        uq_method = self._latent_config["uq"].get("method", "unoise")
        if uq_method == "mixedq" and offset_heuristic:
            offset_heuristic = False      # reference :70-76
        self._offset_heuristic = offset_heuristic
        self.itinf = False
        self._optimizer_config = dict(optimizer_config)
        self._transform_config = transform_config
        self._profile = profile
        self._prior_num_filters = tuple(prior_num_filters)
        if precision not in ("fp32", "bf16x3"):
            raise ValueError(f"precision must be 'fp32' or 'bf16x3', not {precision!r}")
        self._precision = precision
        self._seed = seed
        self._quality_metrics = quality_metrics     # MS-SSIM at eval (reference :321-331); LPIPS is not vendored
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._step = 0
        self._timing = {}
        self._prior = None
        self._init_transforms(transform_config)

    # -- construction (reference :111-149) ---------------------------------------------------------
    def _build_named(self, cfg, cin, seed_offset):
        cfg = dict(cfg)
        t = transform_builder.build(cfg.pop("cls"), **cfg)
        t._precision = self._precision
        t._seed = self._seed + seed_offset
        t._cin = t._cin or cin
        return t

    def _init_transforms(self, transform_config=EMPTY_DICT):
        self._analysis = self._build_named(transform_config["analysis"], 3, 0)
        self._bottleneck_size = b = self._analysis.out_channels(3)
        self._synthesis = self._build_named(transform_config["synthesis"], b, 1)
        ha = transform_config.get("hyper_analysis", dict(cls="HyperAnalysis", bottleneck_size=b))
        hs = transform_config.get("hyper_synthesis", dict(cls="HyperSynthesis", bottleneck_size=b))
        self._hyper_analysis = self._build_named(ha, b, 2)
        self._hyper_bottleneck_size = hb = self._hyper_analysis.out_channels(b)
        self._hyper_synthesis = self._build_named(hs, hb, 3)
        if self._hyper_synthesis.out_channels(hb) != 2 * b:
            raise ValueError("hyper-synthesis must emit 2 * bottleneck channels (mean and scale)")
        self._prior_weights = deep_factorized_init(hb, self._prior_num_filters, seed=self._seed + 4)
        # downsample_factor = 64 / spatial size of the hyper-latents of a 64 x 64 dummy image (:137-140)
        self.downsample_factor = self._compute_downsample_factor()

    def _transforms(self):
        return OrderedDict(analysis=self._analysis, synthesis=self._synthesis,
                           hyper_analysis=self._hyper_analysis, hyper_synthesis=self._hyper_synthesis)

    def _compute_downsample_factor(self):
        for t in self._transforms().values():
            t.build(device=self.device)
        # the reference pushes a zero 64 x 64 image through analysis + hyper-analysis; the spatial size of
        # the result is all it uses, so shape inference suffices
        _, dim = self._hyper_analysis.out_hw(*self._analysis.out_hw(DUMMY_IMG_DIM, DUMMY_IMG_DIM))
        factor = int(DUMMY_IMG_DIM / dim)
        assert dim * factor == DUMMY_IMG_DIM, "Downsample factor should divide evenly into the dummy image size."
        return factor

    # -- weights --------------------------------------------------------------------------------
    def get_weights(self):
        """Flat ``{prefix/name: ndarray}`` over all transforms + the hyper-prior ('prior/...')."""
        out = OrderedDict()
        for pre, t in self._transforms().items():
            for k, v in t.get_weights().items():
                out[f"{pre}/{k}"] = v
        out.update(self._prior_weights)
        return out

    def set_weights(self, weights, _from_trainer=False):
        """Load variables (``get_weights()`` naming).  A ``Trainer`` attached to this model keeps its own flat copy of the
        variables and the Adam moments: weights set from anywhere else make that state stale, so it is dropped and the
        next ``train_step`` builds a fresh one from these weights and ``self._step``."""
        if not _from_trainer:
            self.trainer = None
        for pre, t in self._transforms().items():
            sub = {k[len(pre) + 1:]: v for k, v in weights.items() if k.startswith(pre + "/")}
            t.set_weights(sub)
            t.build(device=self.device)
        shapes = deep_factorized_shapes(self._prior_channels(), self._prior_num_filters)
        pw = OrderedDict()
        for k, shp in shapes.items():
            a = np.asarray(weights[k], np.float32)
            if tuple(a.shape) != tuple(shp):
                raise ValueError(f"{k}: expected {shp}, got {a.shape}")
            pw[k] = a
        self._prior_weights = pw
        self._prior = None
        self._codec = None

    def _prior_channels(self):
        return self._hyper_bottleneck_size

    def _get_prior(self):
        if self._prior is None:
            nl = len(self._prior_num_filters) + 1
            pw = self._prior_weights
            with torch.cuda.device(self.device):
                self._prior = ops.DeepFactorizedPrior([pw[f"prior/matrix_{k}"] for k in range(nl)],
                                                      [pw[f"prior/bias_{k}"] for k in range(nl)],
                                                      [pw[f"prior/factor_{k}"] for k in range(nl - 1)])
        return self._prior

    # -- schedules (reference :151-209) ----------------------------------------------------------
    @property
    def global_step(self):
        return self._itinf_step if self.itinf else self._step

    @property
    def _scheduled_lr(self):
        return compression_lr(self._optimizer_config, self._scheduled_num_steps, self.global_step)

    @property
    def _scheduled_rd_lambda(self):
        if self._rd_lambda <= 0.01 and not self.itinf:
            boundary = int(self._scheduled_num_steps * HIGHER_LAMBDA_UNTIL)
            return self._rd_lambda * (HIGHER_LAMBDA_FACTOR if self.global_step < boundary else 1.0)
        return self._rd_lambda

    @property
    def latent_config(self):
        config = {k: dict(v) if isinstance(v, dict) else v for k, v in self._latent_config.items()}
        cfg = config.get("uq")
        if cfg and cfg.get("method") == "sga":
            cfg["tau"] = sga_schedule_at_step(self.global_step, r=cfg["tau_r"], ub=cfg["tau_ub"],
                                              lb=cfg.get("tau_lb", 1e-8), t0=cfg["tau_t0"])
        return config

    # -- inference path (reference :212-232) -----------------------------------------------------
    def _as_device_images(self, x):
        if isinstance(x, np.ndarray):
            if x.dtype == np.uint8:                                   # raw pixels: scale as data_lib.py:24-25 does
                x = data_lib.normalize_image(x)
            x = ops.to_device(x, self.device)
        if x.dim() == 3:
            x = x.unsqueeze(0)
        return x.contiguous()

    def _timed(self, name, fn, *args):
        """``profile=True`` (reference :142-149, common/profile_utils.py:62-77): per-transform wall time as Metrics scalars
        ``<name>_time`` in seconds -- here GPU time between two HIP events on the launch stream (the reference's
        perf_counter around a tf.function is documented as inaccurate, README.md:44-45)."""
        if not self._profile:
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args)
        e1.record()
        e1.synchronize()
        self._timing[f"{name}_time"] = e0.elapsed_time(e1) * 1e-3
        return out

    def infer_latent_rvs(self, x):
        x = self._as_device_images(x)
        self._timing = {}
        with torch.cuda.device(self.device):
            xp = image_utils.pad_images(x, self.downsample_factor)
            y = self._timed("analysis", self._analysis, xp)
            z = self._timed("hyper_analysis", self._hyper_analysis, y)
        return LatentRVCollection(uq=(UQLatentRV(z), UQLatentRV(y)))

    # -- generative path + losses (reference :234-359, training=False branch) ------------------------
    def _rate_and_reconstruction(self, latent_rvs, want_symbols=False):
        z, y = latent_rvs.uq[0].loc, latent_rvs.uq[1].loc
        z_hat, bits_z = self._get_prior()(z)                          # :254-259 (offset 0)
        hyper = self._timed("hyper_synthesis", self._hyper_synthesis, z_hat)    # :273; split + exp fused below
        y_hat, bits_y, sym = ops.entropy_scale_normal(y, hyper, want_symbols)   # :274-279
        recon = self._timed("synthesis", self._synthesis, y_hat)                # :297
        return dict(z_hat=z_hat, y_hat=y_hat, symbols=sym, hyper=hyper, bits_z=bits_z, bits_y=bits_y, recon=recon)

    def frame_loss_given_latent_rvs(self, image_batch, latent_rvs, training, noise=None, seed=None):
        """reference :234-359.  ``training=False``: hard rounding, uint8 distortion, MS-SSIM.  ``training=True``: the loss value
        the reference's ``train_step`` / ``itinf_train_step`` differentiate -- the latents are perturbed as
        ``latent_config['uq']['method']`` says ('unoise' / 'mixedq': additive uniform noise, :253-259,277-283; anything else,
        e.g. 'sga' / 'soft_round': explicit ``UQLatentRV.sample``, :260-268,285-291), the rate is the noisy densities' at those
        samples and the distortion is taken on unrounded 0-255 floats (data_lib.py:48-52).  Forward value only: the
        gradients live in ``train_step`` (shallow_ntc_amd.train.Trainer) and ``itinf_train_step`` (sga.SGAEngine), which
        draw the same numbers from the same (seed, step).  ``noise`` = (for z, for y) fixes the draw (tests)."""
        if training:
            return self._training_frame(image_batch, latent_rvs, noise, seed)
        return self._finish_frame(self._launch_frame(image_batch, latent_rvs))

    def _training_samples(self, latent_rvs, noise, seed):
        """-> (bits_z[n] or None, bits_y[n], the tensor the synthesis decodes): the training=True branch up to the synthesis."""
        uq = self._latent_config["uq"].get("method", "unoise")
        step = self.global_step
        nz, ny = (None, None) if noise is None else noise
        z_rv, y_rv = latent_rvs.uq
        prior = self._get_prior()
        if uq in ("unoise", "mixedq"):
            z_t = z_rv.sample(True, "unoise", noise=nz, seed=seed, step=2 * step)                 # :253-259
            bits_z, _ = ops.noisy_factorized(prior, z_t)
            z_dec = z_rv.quantize() if uq == "mixedq" else z_t                                    # offset 0
            hyper = self._hyper_synthesis(z_dec)                                                  # :273
            y_t = y_rv.sample(True, "unoise", noise=ny, seed=seed, step=2 * step + 1)             # :277-283
            bits_y, _, _ = ops.noisy_normal(y_t, hyper)
            y_dec = ops.entropy_scale_normal(y_rv.loc, hyper)[0] if uq == "mixedq" else y_t
            return bits_z, bits_y, y_dec
        cfg = dict(self.latent_config["uq"])                                                      # explicit sampling, :260-268
        if uq == "sga":       # the fused kernels of itinf_train_step: same draw, same rate arithmetic
            z_t, _, _, bits_z = ops.sga_factorized_fwd(prior, z_rv.loc, cfg["tau"], nz, seed, step)
            hyper = self._hyper_synthesis(z_t)
            y_t, _, _, _, bits_y = ops.sga_normal_fwd(y_rv.loc, hyper, cfg["tau"], ny, seed, step)
            return bits_z, bits_y, y_t
        z_t = z_rv.sample(True, offset=None, noise=nz, seed=seed, step=2 * step, **cfg)
        bits_z, _ = ops.noisy_factorized(prior, z_t)
        hyper = self._hyper_synthesis(z_t)
        c = y_rv.loc.shape[-1]
        y_t = y_rv.sample(True, offset=hyper[..., :c], noise=ny, seed=seed, step=2 * step + 1, **cfg)       # :285-291
        bits_y, _, _ = ops.noisy_normal(y_t, hyper)
        return bits_z, bits_y, y_t

    def _training_frame(self, image_batch, latent_rvs, noise=None, seed=None):
        x = self._as_device_images(image_batch)
        seed = self._seed if seed is None else seed
        with torch.cuda.device(self.device):
            bits_z, bits_y, y_dec = self._training_samples(latent_rvs, noise, seed)
            recon = self._synthesis(y_dec, training=True)                                         # :297
            sse = ops.float_sse(x, recon)                                                         # unpad + 0-255 floats, unrounded
            rows = [bits_y if bits_z is None else bits_z, bits_y, sse]
            host = torch.stack(rows).cpu().numpy()
            ops.check_conv_status()
        rd_loss, metrics = self._finish_metrics(x.shape, None if bits_z is None else host[0], host[1], host[2])
        metrics.record_image("reconstruction", recon)
        return rd_loss, metrics

    def _launch_frame(self, image_batch, latent_rvs=None):
        """Everything of end_to_end_frame_loss(training=False) that runs on the GPU, launched on the current stream with
        no host synchronisation: -> the pending device results for ``_finish_frame``."""
        x = self._as_device_images(image_batch)
        with torch.cuda.device(self.device):
            if latent_rvs is None:
                latent_rvs = self.infer_latent_rvs(x)
            r = self._rate_and_reconstruction(latent_rvs)
            sse, _ = ops.pixels_sse(x, r["recon"])                    # unpad + floats_to_pixels + mse fused
            dev = torch.stack([r["bits_z"], r["bits_y"], sse.to(torch.float64)])
            quality = None
            h, w = x.shape[1], x.shape[2]
            if self._msssim_applies(h, w):
                quality = ops.image_quality_launch(ops.pixels_float(x, h, w), ops.pixels_float(r["recon"], h, w), 255.0)
        return dict(shape=tuple(x.shape), dev=dev, quality=quality, recon=r["recon"])

    def _finish_frame(self, pending):
        """Host side of a launched frame: one device -> host copy, then the reference's float32 metric arithmetic."""
        host = pending["dev"].cpu().numpy()
        ops.check_conv_status()                      # the copy above synchronised the stream: a flagged stream-K launch raises here
        msssim = None
        if pending["quality"] is not None:
            sums, counts, single = pending["quality"]
            msssim = ops.image_quality_finish(sums.cpu().numpy(), counts, single)
        rd_loss, metrics = self._finish_metrics(pending["shape"], host[0], host[1], host[2], msssim)
        metrics.record_image("reconstruction", pending["recon"])
        return rd_loss, metrics

    def _msssim_applies(self, h, w):
        """reference :321-331: (MS-)SSIM of the uint8-quantised images whenever TensorFlow itself can compute it."""
        if not self._quality_metrics or min(h, w) < 11:
            return False
        # the reference switches to ssim_multiscale once either side reaches 160 (:325-329), which TensorFlow itself
        # rejects when the fifth scale is smaller than the 11 x 11 window; report no MS-SSIM instead of failing the step
        return not ((h >= 160 or w >= 160) and min(h, w) < 11 * 16)

    def _msssim(self, x, recon):
        """Per-image (MS-)SSIM of the uint8-quantised images (reference :321-331), or None."""
        h, w = x.shape[1], x.shape[2]
        if not self._msssim_applies(h, w):
            return None
        return ops.image_quality(ops.pixels_float(x, h, w), ops.pixels_float(recon, h, w), 255.0)

    def _finish_metrics(self, x_shape, bits_z, bits_y, sse, msssim=None, sched=None):
        """``sched`` = (scheduled_lr, sched_rd_lambda, tau) of the step the numbers belong to, when that is not the current one
        (metrics of an SGA step fetched later)."""
        n, h, w, c = x_shape
        num_pixels = np.float32(h * w)                                                  # :302
        bits_z = None if bits_z is None else bits_z.astype(np.float32)
        bits_y = bits_y.astype(np.float32)
        hyper_bpp = np.float32(0.0) if bits_z is None else np.float32(bits_z.mean(dtype=np.float32) / num_pixels)
        latent_bpp = np.float32(bits_y.mean(dtype=np.float32) / num_pixels)           # :306-307
        for name, v in (("hyper_latent_bpp", hyper_bpp), ("latent_bpp", latent_bpp)):
            if not np.isfinite(v):                                                      # check_numerics :308-309
                raise capi.NonFiniteError(capi.ERR_NONFINITE, f"{name} : Tensor had NaN/Inf values")
        bpp = np.float32(hyper_bpp + latent_bpp)
        mses, psnrs = image_utils.mse_psnr_from_sse(sse, h * w * c)                     # :315
        mse, psnr = np.float32(mses.mean(dtype=np.float32)), np.float32(psnrs.mean(dtype=np.float32))
        lam = self._scheduled_rd_lambda if sched is None else sched[1]
        rd_loss = np.float32(bpp + np.float32(lam) * mse)                               # :343
        if not np.isfinite(rd_loss):                                                    # :356
            raise capi.NonFiniteError(capi.ERR_NONFINITE, "rd_loss : Tensor had NaN/Inf values")
        metrics = Metrics.make()
        metrics.record_scalar("sched_rd_lambda", lam)
        if self.latent_config["uq"].get("method") == "sga":
            metrics.record_scalar("tau", self.latent_config["uq"]["tau"] if sched is None else sched[2])
        if msssim is not None:                                                          # :321-331
            ms = np.float32(np.asarray(msssim, np.float32).mean(dtype=np.float32))
            with np.errstate(divide="ignore"):
                db = np.float32((-10.0 * np.log10(1.0 - np.asarray(msssim, np.float64))).mean())
            metrics.record_scalars(dict(msssim=float(ms), msssim_db=float(db)))
        metrics.record_scalars(dict(rd_loss=float(rd_loss), bpp=float(bpp), mse=float(mse), psnr=float(psnr),
                                    scheduled_lr=self._scheduled_lr if sched is None else sched[0]))
        if self._profile:                                                               # :350-351
            metrics.record_scalars(dict(getattr(self, "_timing", {})))
        return float(rd_loss), metrics

    def end_to_end_frame_loss(self, image_batch, training):
        latent_rvs = self.infer_latent_rvs(image_batch)
        return self.frame_loss_given_latent_rvs(image_batch, latent_rvs=latent_rvs, training=training)

    def validation_step(self, image_batch, training=False) -> Metrics:
        _, metrics = self.end_to_end_frame_loss(image_batch, training=training)
        return metrics

    def evaluate(self, images, lookahead=3, group=8):
        """Reference :415-433: a [B,H,W,3] tensor is evaluated one [1,H,W,3] image at a time, an
        iterable is taken as is; yields one Metrics per image, in order.

        Same numbers as the reference's loop, image by image and in order, but not its stalls.  One image alone leaves most
        of the GPU idle (a 512 x 768 image is 72 ... 288 workgroups per layer on 256 CUs), so up to ``group`` images OF THE
        SAME SHAPE among the next ``lookahead * group`` are launched as one batch -- every kernel on this path gives an
        image bit-identical results alone or inside any batch -- and up to ``lookahead`` such launches are
        in flight on round-robin HIP streams before the oldest one's results are copied to the host, so the device never
        waits for Python.  lookahead=1 is the strictly serial one-image-per-pass reference behaviour.  (Defaults from
        tools/time_evaluate.py on the Kodak-shaped set: 4 x 4 178, 2 x 8 189, 3 x 8 204 Mpixel/s untuned; the launches of eight
        images fill the quarter-resolution layers' 8 x 32 tiles for one and a half rounds of the device instead of three quarters of one.)"""
        tensor_input = isinstance(images, (torch.Tensor, np.ndarray))
        if tensor_input:
            images = [images[i:i + 1] for i in range(images.shape[0])]
        lookahead, group = max(1, int(lookahead)), max(1, int(group))
        if lookahead == 1:
            for img in images:
                _, metrics = self.end_to_end_frame_loss(img, training=False)
                yield metrics
            return
        if self._profile:
            group = 1                                                 # per-transform times are per pass
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            self._eval_streams = ops.side_streams(lookahead, self.device)      # the library's one pool: a stream per hardware queue
            window = []                  # entries in input order: [x, metrics or None, launched]
            inflight = []                # (stream, pending, entries)
            it = iter(images)
            k = 0
            done = False
            while True:
                while not done and len(window) < lookahead * group:
                    try:
                        window.append([self._as_device_images(next(it)), None, False])
                    except StopIteration:
                        done = True
                while len(inflight) < lookahead:
                    first = next((e for e in window if not e[2]), None)
                    if first is None:
                        break
                    members = [e for e in window if not e[2] and e[0].shape == first[0].shape][:group]
                    if first[0].shape[0] != 1:
                        members = [first]                             # an iterable of batches: each is taken as it is
                    for e in members:
                        e[2] = True
                    st = self._eval_streams[k % lookahead]
                    k += 1
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        x = members[0][0] if len(members) == 1 else torch.cat([e[0] for e in members])
                        inflight.append((st, self._launch_frame(x), members))
                if not inflight:
                    break
                st, pending, members = inflight.pop(0)
                with torch.cuda.stream(st):
                    for e, (_, metrics) in zip(members, self._finish_frames(pending, len(members))):
                        e[1] = metrics
                while window and window[0][1] is not None:
                    yield window.pop(0)[1]
            for st in self._eval_streams:
                cur.wait_stream(st)

    def _finish_frames(self, pending, count):
        """``_finish_frame`` for a launch that holds ``count`` one-image frames: one host copy, then each image's own
        metrics exactly as if it had been launched alone."""
        if count == 1:
            return [self._finish_frame(pending)]
        host = pending["dev"].cpu().numpy()
        ops.check_conv_status()
        msssim = None
        if pending["quality"] is not None:
            sums, counts, single = pending["quality"]
            msssim = ops.image_quality_finish(sums.cpu().numpy(), counts, single)
        out = []
        for i in range(count):
            rd_loss, metrics = self._finish_metrics((1,) + tuple(pending["shape"][1:]), host[0][i:i + 1], host[1][i:i + 1],
                                                    host[2][i:i + 1], None if msssim is None else msssim[i:i + 1])
            metrics.record_image("reconstruction", pending["recon"][i:i + 1])
            out.append((rd_loss, metrics))
        return out

    def evaluate_batched(self, images):
        """Same numbers as ``evaluate`` for same-shaped images, but one launch sequence for the whole
        batch (independent images fill the GPU): returns per-image dicts(bpp, mse, psnr, rd_loss)."""
        x = self._as_device_images(images)
        with torch.cuda.device(self.device):
            r = self._rate_and_reconstruction(self.infer_latent_rvs(x))
            sse, _ = ops.pixels_sse(x, r["recon"])
            host = torch.stack([r["bits_z"], r["bits_y"], sse.to(torch.float64)]).cpu().numpy()
            ops.check_conv_status()
            msssim = self._msssim(x, r["recon"])
        out = []
        for i in range(x.shape[0]):
            _, m = self._finish_metrics((1,) + tuple(x.shape[1:]), host[0][i:i + 1], host[1][i:i + 1], host[2][i:i + 1],
                                        None if msssim is None else msssim[i:i + 1])
            out.append(m.scalars_float)
        return out

    # -- codec regions -------------------------------------------------------------
    def encode(self, x, check=True):
        """x -> (z_hat, symbols int32, bits_z[n], bits_y[n]); needs the hyper-synthesis for mu, sigma.  ``check``: wait for the
        launches and raise if a stream-K hand-off timed out (``ops.check_conv_status``); a caller that keeps several calls in
        flight passes False and checks where it synchronises itself."""
        x = self._as_device_images(x)
        with torch.cuda.device(self.device):
            lat = self.infer_latent_rvs(x)
            z_hat, bits_z = self._get_prior()(lat.uq[0].loc)
            hyper = self._hyper_synthesis(z_hat)
            _, bits_y, sym = ops.entropy_scale_normal(lat.uq[1].loc, hyper, want_symbols=True)
            if check:
                ops.check_conv_status()
        return z_hat, sym, bits_z, bits_y

    def decode(self, z_hat, symbols, image_hw, reference=None, check=True):
        """(z_hat, symbols) -> hyper-synthesis -> y_hat = symbols + mu -> synthesis -> uint8 pixels
        [n, H, W, 3] (and the per-image integer SSE against ``reference`` if given).  ``check`` as in ``encode``."""
        with torch.cuda.device(self.device):
            hyper = self._hyper_synthesis(z_hat)
            if self._synthesis.takes_s3(symbols.shape[1], symbols.shape[2]):      # bf16x3: y_hat leaves the dequantisation pre-split
                y_hat = ops.dequant_split3(symbols, hyper)
            else:
                y_hat = ops.dequant_scale_normal(symbols, hyper)
            out = self._pixels(y_hat, image_hw, reference)
            if check:
                ops.check_conv_status()
            return out

    def decode_set(self, codes, check=True):
        """``decode`` for a SET of batches of different image sizes (the Kodak set's two orientations): ``codes`` =
        [(z_hat, symbols, image_hw[, reference])] -> the list of what ``decode`` returns for each.  The hyper-syntheses and
        dequantisations of the batches run side by side on one stream per batch; a two-layer synthesis then takes ALL batches
        in ONE launch (common/transforms.py ``hidden_many``: per-image geometry inside the kernel), and the output layers
        follow per batch.  Same pixels as one ``decode`` per batch."""
        codes = [tuple(c) + (None,) * (4 - len(c)) for c in codes]
        syn = self._synthesis
        if len(codes) < 2 or len(codes) > 4 or not hasattr(syn, "hidden_many") or self._synthesis.takes_s3(*codes[0][1].shape[1:3]):
            return [self.decode(z, s, hw, reference=r, check=check) for z, s, hw, r in codes]
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            self._set_streams = ops.side_streams(len(codes), self.device)
            y_hats = []
            for st, (z_hat, sym, _hw, _r) in zip(self._set_streams, codes):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    y_hat = ops.dequant_scale_normal(sym, self._hyper_synthesis(z_hat))
                y_hat.record_stream(cur)
                y_hats.append(y_hat)
            for st in self._set_streams[:len(codes)]:
                cur.wait_stream(st)
            hidden = syn.hidden_many(y_hats)
            outs = []
            if hidden is None:
                for y_hat, (_z, _s, hw, ref) in zip(y_hats, codes):
                    outs.append(self._pixels(y_hat, hw, ref))
            else:
                for st, hid, (_z, _s, hw, ref) in zip(self._set_streams, hidden, codes):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        px, sse = syn.pixels_from_hidden(hid, hw[0], hw[1], ref)
                    hid.record_stream(st)
                    px.record_stream(cur)
                    if sse is not None:
                        sse.record_stream(cur)
                    outs.append(px if ref is None else (px, sse))
                for st in self._set_streams[:len(codes)]:
                    cur.wait_stream(st)
            if check:
                ops.check_conv_status()
            return outs

    def _pixels(self, y_hat, image_hw, reference=None):
        """synthesis -> unpad -> floats_to_pixels -> quantize_image (reference :297-317) as uint8 [n, H, W, 3], plus the
        per-image integer SSE when ``reference`` is given.  The two-layer syntheses emit the pixels from their last
        launch; the others go through the float reconstruction."""
        if hasattr(self._synthesis, "forward_pixels"):
            px, sse = self._synthesis.forward_pixels(y_hat, image_hw[0], image_hw[1], reference)
            return px if reference is None else (px, sse)
        recon = self._synthesis(y_hat)
        if reference is None:
            return ops.to_pixels(recon, image_hw[0], image_hw[1])
        sse, px = ops.pixels_sse(reference, recon, want_pixels=True)
        return px, sse

    # -- bitstream (the reference itself only estimates the rate) -----------------------
    def _get_codec(self):
        from ..entropy_coding import Codec
        if getattr(self, "_codec", None) is None:
            self._codec = Codec(self)
        return self._codec

    def compress(self, x) -> bytes:
        """Images -> self-contained bitstream (rANS over the integer CDF tables of both entropy models)."""
        return self._get_codec().compress(x)

    def compress_many(self, xs):
        """Several batches (e.g. one per image size of a set) -> their bitstreams; the batches' launches run side by side and
        the set costs two host synchronisations (entropy_coding.Codec.compress_many).  Same bytes as one ``compress`` per batch."""
        return self._get_codec().compress_many(list(xs))

    def decompress(self, blob: bytes):
        """Bitstream -> uint8 pixels [n, H, W, 3]; bit-identical to ``decode(encode(x))``."""
        return self._get_codec().decompress(blob)

    def decompress_many(self, blobs):
        """Several bitstreams -> their pixel batches; the entropy-decoding launches of all of them run side by side
        (entropy_coding.Codec.decompress_many).  Same pixels as one ``decompress`` per blob."""
        return self._get_codec().decompress_many(list(blobs))

    # -- training (reference :375-383) -----------------------------------------------------------------------
    def train_step(self, image_batch):
        """One optimizer step on ``image_batch`` (tape.gradient of end_to_end_frame_loss(training=True) + Adam,
        reference :375-383); returns Metrics with the reference's scalar keys.  The training state (flat parameter /
        gradient / moment buffers, adjoint plans) lives in ``shallow_ntc_amd.train.Trainer`` and is created on first
        use; ``self.trainer.sync_model()`` loads the trained variables back into the inference path."""
        if getattr(self, "trainer", None) is None:
            from ..train import Trainer
            self.trainer = Trainer(self, seed=self._seed)
        d = self.trainer.train_step(image_batch)
        metrics = Metrics.make()
        metrics.record_scalars({k: d[k] for k in ("rd_loss", "bpp", "mse", "psnr", "scheduled_lr", "sched_rd_lambda")})
        return metrics

    # -- iterative inference (reference :389-413, common/itinf_lib.py:26-93) ----------------------------
    def initialize_itinf(self, image_batch):
        """latent_rvs = trainable copy of the encoder's latents; fresh Adam state (:389-395)."""
        from ..sga import SGAEngine
        if self._optimizer_config.get("global_clipnorm") is not None:
            raise NotImplementedError("gradient clipping is not used by the reference's itinf config")
        self.latent_rvs = self.infer_latent_rvs(image_batch).get_trainable_copy()
        self._sga = getattr(self, "_sga", None) or SGAEngine(self)
        self._adam = [dict(m=torch.zeros_like(rv.loc), v=torch.zeros_like(rv.loc)) for rv in self.latent_rvs.uq]
        self.itinf = True
        self._itinf_step = 0
        self._itinf_pending = None

    @property
    def itinf_trainable_variables(self):
        return self.latent_rvs.trainable_variables

    def itinf_train_step(self, image_batch, noise=None, seed=0, fetch=True):
        """One SGA step: loss = bpp + lambda * MSE(unrounded 0-255 floats), gradients to [z_loc, y_loc] only,
        Keras-Adam update (:397-408).  ``noise`` = (gumbel_z, gumbel_y) makes the step deterministic.
        ``fetch`` False: the step's three scalars stay on the device and nothing synchronises (the reference's step is a
        tf.function whose metrics are only converted where the loop logs them, common/itinf_lib.py:67-75); returns None, and
        ``itinf_last_metrics()`` fetches the most recent step's metrics when the caller wants them."""
        x = self._as_device_images(image_batch)
        cfg = self.latent_config["uq"]
        if cfg.get("method") != "sga":
            raise NotImplementedError("itinf_train_step implements latent_config uq.method == 'sga'")
        tau = cfg["tau"]
        lr = self._scheduled_lr
        locs = [rv.loc for rv in self.latent_rvs.uq]                     # (z_loc, y_loc); the factorized model: (y_loc,)
        with torch.cuda.device(self.device):
            r = self._sga.loss_and_grads(x, locs[0] if len(locs) == 2 else None, locs[-1], tau, self._scheduled_rd_lambda,
                                         step=self._itinf_step, seed=seed,
                                         noise_z=None if noise is None else noise[0],
                                         noise_y=None if noise is None else noise[-1])
            t = self._itinf_step + 1
            grads = [r["g_z"], r["g_y"]] if len(locs) == 2 else [r["g_y"]]
            for p, g, st in zip(locs, grads, self._adam):
                ops.adam_step(p, g, st["m"], st["v"], lr, t, self._optimizer_config.get("beta_1", 0.9),
                              self._optimizer_config.get("beta_2", 0.999), self._optimizer_config.get("epsilon", 1e-7))
            # the metrics depend on (step, lr, lambda, tau) of THIS step: keep them with the device scalars
            self._itinf_pending = dict(dev=torch.stack([r["bits_z"], r["bits_y"], r["sse"]]), shape=tuple(x.shape), two=len(locs) == 2,
                                       scalars=(self._scheduled_lr, self._scheduled_rd_lambda, tau))
        self._itinf_step += 1
        self.last_grads = tuple(grads)
        return self.itinf_last_metrics() if fetch else None

    def itinf_last_metrics(self):
        """Metrics of the most recent ``itinf_train_step`` (one device -> host copy; raises if a stream-K launch since the last
        check was flagged)."""
        pend = getattr(self, "_itinf_pending", None)
        if pend is None:
            raise RuntimeError("no SGA step has run since initialize_itinf")
        with torch.cuda.device(self.device):
            host = pend["dev"].cpu().numpy()
            ops.check_conv_status()
        _, metrics = self._finish_metrics(pend["shape"], host[0] if pend["two"] else None, host[1], host[2], sched=pend["scalars"])
        return metrics

    def itinf_validation_step(self, image_batch, training=False) -> Metrics:
        _, metrics = self.frame_loss_given_latent_rvs(image_batch, latent_rvs=self.latent_rvs, training=training)
        return metrics

"""Pixel-domain helpers (reference common/image_utils.py:22-71) on CUDA NHWC float32 tensors."""
from __future__ import annotations

import math

import numpy as np

from .. import ops


def pad_images(x, div: int, padding_mode="reflect"):
    """Reflect-pad bottom/right so H and W are divisible by ``div`` (image_utils.py:41-66)."""
    if padding_mode != "reflect":
        raise NotImplementedError(padding_mode)
    h, w = x.shape[1], x.shape[2]
    hp, wp = -(-h // div) * div, -(-w // div) * div
    return ops.pad_reflect(x, hp, wp)


def unpad_images(x, unpadded_shape):
    """image_utils.py:69-71."""
    return ops.crop(x, unpadded_shape[1], unpadded_shape[2])


def mse_psnr_from_sse(sse, num_values, max_val=255.0):
    """image_utils.py:26-38 given per-image sums of squared differences (host arrays):
    mse = sse / (H*W*C); psnr = -10 (ln mse - 2 ln max) / ln 10, in float32 like the reference."""
    mses = (np.asarray(sse, np.float64) / float(num_values)).astype(np.float32)
    with np.errstate(divide="ignore"):
        psnrs = (np.float32(-10.0) * (np.log(mses) - np.float32(2.0) * np.log(np.float32(max_val)))
                 / np.log(np.float32(10.0))).astype(np.float32)
    return mses, psnrs


def psnr_from_mse(mse, max_val=255.0):
    return -10.0 * (math.log(mse) - 2.0 * math.log(max_val)) / math.log(10.0)

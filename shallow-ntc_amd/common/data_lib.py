"""Image scaling helpers (reference common/data_lib.py:24-52).  Host-side: images enter the device
already normalised; the inverse mapping + uint8 quantisation is fused into ops.pixels_sse."""
import numpy as np


def normalize_image(image):
    """uint8/float pixels in [0,255] -> float32 in [-0.5, 0.5] (data_lib.py:24-25), computed in float32."""
    return (np.asarray(image, np.float32) / np.float32(255.0) - np.float32(0.5)).astype(np.float32)


def unnormalize_image(x):
    return (np.asarray(x, np.float32) + np.float32(0.5)) * np.float32(255.0)


def synthetic_images(n, h, w, seed=1234):
    """Seeded smooth test images: 16 random low-frequency cosines per channel +
    N(0, 4^2) noise, clipped to uint8.  Returns uint8 [n,h,w,3]."""
    rng = np.random.default_rng(seed)
    yy = np.arange(h, dtype=np.float32)[:, None] / max(h, 1)
    xx = np.arange(w, dtype=np.float32)[None, :] / max(w, 1)
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        for c in range(3):
            img = np.full((h, w), 128.0, np.float32)
            for _ in range(16):
                fy, fx = rng.uniform(0, 6, size=2)
                ph = rng.uniform(0, 2 * np.pi)
                amp = rng.uniform(4, 24)
                img += amp * np.cos(2 * np.pi * (fy * yy + fx * xx) + ph).astype(np.float32)
            img += rng.normal(0, 4.0, size=(h, w)).astype(np.float32)
            out[i, :, :, c] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out

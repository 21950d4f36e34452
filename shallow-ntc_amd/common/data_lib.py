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


# ---- input pipeline from image files (reference :32-46, :86-109, :113-143) --------------------------------------
def read_png(path):
    """uint8 [h, w, 3] (reference read_png -> tf.image.decode_image(channels=3))."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), np.uint8)


def process_image(image, crop=None, patchsize=None, normalize=True, rng=None):
    """reference process_image (:32-46): optional random / center crop to patchsize x patchsize, float32, normalise."""
    if crop is not None:
        assert patchsize and patchsize > 0
        h, w = image.shape[:2]
        if h < patchsize or w < patchsize:
            raise ValueError(f"image {h} x {w} smaller than the {patchsize} patch")
        if crop == "random":
            rng = rng or np.random.default_rng()
            top, left = int(rng.integers(0, h - patchsize + 1)), int(rng.integers(0, w - patchsize + 1))
        elif crop == "center":                                  # image_utils.center_crop_image
            top, left = (h - patchsize) // 2, (w - patchsize) // 2
        else:
            raise NotImplementedError(crop)
        image = image[top:top + patchsize, left:left + patchsize]
    image = np.asarray(image, np.float32)
    return normalize_image(image) if normalize else image


def get_dataset_from_glob(file_glob, shuffle, repeat, drop_remainder, batchsize, crop=None, patchsize=None, normalize=True,
                          seed=0):
    """Generator of NHWC float32 batches from PNG (or any PIL-readable) files, the semantics of reference :86-109:
    sorted file list, optional reshuffle per epoch, repeat, batches of same-shaped images (full-size evaluation uses
    batchsize 1, as the reference's val_data_config does)."""
    import glob
    files = sorted(glob.glob(file_glob))
    if not files:
        raise RuntimeError(f"No images found with glob '{file_glob}'.")
    rng = np.random.default_rng(seed)

    def gen():
        while True:
            order = list(rng.permutation(len(files))) if shuffle else list(range(len(files)))
            batch = []
            for i in order:
                batch.append(process_image(read_png(files[i]), crop, patchsize, normalize, rng))
                if len(batch) == batchsize:
                    yield np.stack(batch)
                    batch = []
            if batch and not drop_remainder:
                yield np.stack(batch)
            if not repeat:
                return

    return gen()


def get_dataset(file_glob, split, batchsize, patchsize, normalize=True, seed=0):
    """reference get_dataset (:113-143) for file-glob datasets: train = shuffled, repeated, random crops, full batches;
    otherwise one ordered pass with center crops (or full images when patchsize is None)."""
    train = split == "train"
    crop = ("random" if train else "center") if patchsize is not None else None
    return get_dataset_from_glob(file_glob, train, train, train, batchsize, crop, patchsize, normalize, seed)

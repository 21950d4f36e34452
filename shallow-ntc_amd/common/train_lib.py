"""Metrics container (reference common/train_lib.py:22-76); the training loop itself is out of scope."""
from typing import Any, Mapping, NamedTuple


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

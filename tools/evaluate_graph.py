#!/usr/bin/env python3
"""One image at a time (the reference's evaluate() loop): the per-image launch sequence eager against replayed from a HIP graph
captured once per image shape.  python tools/evaluate_graph.py [passes]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd import ops
from shallow_ntc_amd.common import data_lib
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.CONFIGS["two_layer_syn"]())
model._quality_metrics = False
images = [torch.from_numpy(data_lib.normalize_image(data_lib.synthetic_images(1, 512, 768, seed=i))).to(dev) for i in range(8)]
with ops.autotune():
    list(model.evaluate(images[:2], lookahead=1))


def serial_eager():
    return [m.scalars_float for m in model.evaluate(images, lookahead=1)]


want = serial_eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    serial_eager()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / (passes * len(images))

with torch.cuda.device(dev):
    xs = images[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            model._launch_frame(xs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pending = model._launch_frame(xs)


def serial_graph():
    out = []
    for img in images:
        xs.copy_(img)
        g.replay()
        _, m = model._finish_frame(pending)
        out.append(m.scalars_float)
    return out


got = serial_graph()
assert got == want, (got[0], want[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    serial_graph()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / (passes * len(images))
px = 512 * 768 / 1e6
print(f"serial evaluate, 512x768: eager {1e3 * te:.3f} ms per image ({px / te:.1f} Mpixel/s) | graph replay {1e3 * tg:.3f} ms ({px / tg:.1f} Mpixel/s) | x{te / tg:.3f}; same metrics")

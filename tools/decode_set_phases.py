"""Where Model.decode_set spends its time on the Kodak-shaped set: the two batch shapes' hyper-syntheses + dequantisations on two
streams, ONE synthesis launch for all 24 images, the output layers -- each phase between HIP events, next to the default
orchestration (one Model.decode per batch shape on its own stream).  python tools/decode_set_phases.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from shallow_ntc_amd import ops  # noqa: E402
from shallow_ntc_amd.mshyper import configs  # noqa: E402
from shallow_ntc_amd.mshyper.models import Model  # noqa: E402

dev = torch.device("cuda:0")
model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.005))
g = torch.Generator(device=dev)
g.manual_seed(99)
codes = []
for n, (h, w) in ((18, (512, 768)), (6, (768, 512))):
    z_hat = torch.round(3.0 * torch.randn((n, h // 64, w // 64, 320), device=dev, generator=g)).contiguous()
    u = torch.rand((n, h // 16, w // 16, 320), device=dev, generator=g) - 0.5
    sym = torch.round(-2.0 * torch.sign(u) * torch.log1p(-2.0 * u.abs())).to(torch.int32).contiguous()
    codes.append((z_hat, sym, (h, w)))
side = [torch.cuda.Stream(device=dev) for _ in codes]
syn = model._synthesis


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def two_streams():
    cur = torch.cuda.current_stream()
    for st, (z, s, hw) in zip(side, codes):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            model.decode(z, s, hw, check=False)
    for st in side:
        cur.wait_stream(st)


def set_phases(record):
    cur = torch.cuda.current_stream()
    t0 = ev()
    ys = []
    for st, (z, s, hw) in zip(side, codes):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            ys.append(ops.dequant_scale_normal(s, model._hyper_synthesis(z)))
    for st in side:
        cur.wait_stream(st)
    t1 = ev()
    hid = syn.hidden_many(ys)
    t2 = ev()
    for st, hd, (_z, _s, hw) in zip(side, hid, codes):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            syn.pixels_from_hidden(hd, hw[0], hw[1])
    for st in side:
        cur.wait_stream(st)
    t3 = ev()
    record.append((t0, t1, t2, t3))


for fn in (two_streams, lambda: model.decode_set(codes, check=False)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a = ev()
    for _ in range(50):
        fn()
    b = ev()
    torch.cuda.synchronize()
    print("two streams, one decode per batch shape" if fn is two_streams else "Model.decode_set", f"{a.elapsed_time(b) / 50:.4f} ms per step", flush=True)
rec = []
for _ in range(10):
    set_phases([])
torch.cuda.synchronize()
for _ in range(30):
    set_phases(rec)
torch.cuda.synchronize()
ph = np.array([[t[i].elapsed_time(t[i + 1]) for i in range(3)] for t in rec])
print("decode_set phases (median ms): hyper-syntheses + dequantisation on two streams %.4f | synthesis, one launch %.4f | output layers %.4f | sum %.4f"
      % (*np.median(ph, axis=0), np.median(ph.sum(axis=1))))
# the hyper-synthesis phase alone, serial
for _ in range(5):
    for z, s, hw in codes:
        model._hyper_synthesis(z)
torch.cuda.synchronize()
a = ev()
for _ in range(30):
    for z, s, hw in codes:
        model._hyper_synthesis(z)
b = ev()
torch.cuda.synchronize()
print(f"both hyper-syntheses on ONE stream: {a.elapsed_time(b) / 30:.4f} ms")

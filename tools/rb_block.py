"""Time the whole-ResidualBlock launch (csrc/rb_fused.hip) against the layers it replaces (1x1 head + fused 3x3 / 1x1 tail),
bursts of launches between HIP events, interleaved rounds in one process.  python tools/rb_block.py [n h w]..."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from shallow_ntc_amd import _capi as capi  # noqa: E402
from shallow_ntc_amd import ops  # noqa: E402


def burst(fn, reps=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    shapes = [(18, 256, 384), (18, 128, 192), (18, 64, 96), (6, 384, 256), (64, 128, 128), (64, 64, 64), (1, 256, 384), (1, 128, 192)]
    if len(sys.argv) > 3:
        v = [int(a) for a in sys.argv[1:]]
        shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    c = 192
    mk = lambda scale, *shape: torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).to(dev)
    w0, b0, w1, b1, w2, b2 = mk(0.08, 1, 1, c, c // 2), mk(1, c // 2), mk(0.05, 3, 3, c // 2, c // 2), mk(1, c // 2), mk(0.1, 1, 1, c // 2, c), mk(1, c)
    la = ops.ConvPlan("conv", w0, b0, 1, "relu")
    lb = ops.ConvPlan("conv", w1, b1, 1, "relu")
    lc = ops.ConvPlan("conv", w2, b2, 1, None, capi.PRO_NONE, capi.EPI_ADD)
    block = ops.ResBlockPlan(w0, b0, w1, b1, w2, b2)
    split = ops.ResBlockPlan(w0, b0, w1, b1, w2, b2, precision="bf16x3")
    for n, h, w in shapes:
        x = mk(1.0, n, h, w, c)
        three = lambda: lb.fused(lc, la(x), res=x) if lb.fusable_with(lc) else lc(lb(la(x)), res=x)
        one = lambda: block(x)
        same = torch.equal(one(), three())
        gf = block.flops(n, h, w) / 1e9
        t3, t1, ts = [], [], []
        for _ in range(5):
            t3.append(burst(three))
            t1.append(burst(one))
            ts.append(burst(lambda: split(x)))
        m3, m1, ms = float(np.median(t3)), float(np.median(t1)), float(np.median(ts))
        print(f"{n:3d}x{h}x{w}: layers {m3:.4f} ms {gf / m3:.1f} TF | block {m1:.4f} ms {gf / m1:.1f} TF (min {min(t1):.4f}) | "
              f"x{m3 / m1:.3f} | bit-identical {same} | bf16x3 block {ms:.4f} ms {gf / ms:.1f} TF-eq x{m1 / ms:.2f} | "
              f"tiles {ops.ResBlockPlan.tiles(n, h, w)}", flush=True)


if __name__ == "__main__":
    main()

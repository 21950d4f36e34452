"""Time the fused first layer of the two-layer synthesis (csrc/syn_fused.hip: transposed convolution + activation + residual in
one launch) against the launches it replaces (phase-grouped gather GEMM, then the tail kernel's stage 1 as sntc_two_layer_hidden),
bursts of launches between HIP events, interleaved rounds in one process.
python tools/syn_block.py [--ch 12] [--res 1] [n h w]...      (h, w: latent size; "n h w + n h w" = one ragged call)"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--ch", type=int, default=12)
    ap.add_argument("--res", type=int, default=1)
    ap.add_argument("--cin", type=int, default=320)
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("shapes", nargs="*")
    args = ap.parse_args()
    sets = [[(18, 32, 48)], [(6, 48, 32)], [(18, 32, 48), (6, 48, 32)], [(64, 16, 16)], [(5, 76, 76)], [(1, 32, 48)]]
    if args.shapes:
        sets, cur, vals = [], [], []
        for tok in args.shapes + ["+"]:
            if tok == "+" or len(vals) == 3:
                if vals:
                    cur.append(tuple(vals))
                    vals = []
                if tok == "+":
                    continue
            vals.append(int(tok))
        sets = [cur]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    ch, res, cin = args.ch, bool(args.res), args.cin
    cp = ch * (2 if res else 1)
    mk = lambda scale, *shape: torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).to(dev)
    w1, b1 = mk(0.5 / np.sqrt(cin), 13, 13, cp, cin), mk(0.1, cp)
    beta = torch.from_numpy((1.0 + rng.random(ch)).astype(np.float32)).to(dev)
    gamma = torch.from_numpy((0.1 * np.eye(ch) + 0.02 * rng.random((ch, ch))).astype(np.float32)).to(dev)
    up = ops.ConvPlan("convT", w1, b1, 8)
    syn = ops.SynPlan(w1, b1, 8, ch, res, 1, beta, gamma)
    if args.workgroups:
        syn.set_workgroups(args.workgroups)
    print("units (shifts per slab, phases, tile steps per slab):", [(u[0], u[1], u[2]) for u in syn.units()],
          "tile steps per slab in all:", sum(u[2] for u in syn.units()), flush=True)
    for shapes in sets:
        xs = [mk(1.0, n, h, w, cin) for n, h, w in shapes]
        layers = lambda: [ops.two_layer_hidden(up(x), ch, res, 1, beta, gamma) for x in xs]
        conv_only = lambda: [up(x) for x in xs]
        one = lambda: syn(xs)
        same = all(torch.equal(a, b) for a, b in zip(one(), layers()))
        px = sum(n * h * w for n, h, w in shapes)
        gf = syn.flops(px) / 1e9
        tl, tc, t1 = [], [], []
        for _ in range(5):
            tl.append(burst(layers))
            tc.append(burst(conv_only))
            t1.append(burst(one))
        ml, mc, m1 = float(np.median(tl)), float(np.median(tc)), float(np.median(t1))
        print(f"{'+'.join(f'{n}x{h}x{w}' for n, h, w in shapes)}: layers {ml:.4f} ms (conv alone {mc:.4f} ms {gf / mc:.1f} TF) | "
              f"fused {m1:.4f} ms {gf / m1:.1f} TF (min {min(t1):.4f}) | x{ml / m1:.3f} | bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()

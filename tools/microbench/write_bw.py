#!/usr/bin/env python3
"""What a pure STORE stream reaches on this device (the roof of csrc/rgb_conv.hip, whose output is 64 x its input): torch fill /
zero / copy of buffers of the first layer's output size, bursts between HIP events.  python tools/microbench/write_bw.py"""
import numpy as np
import torch

dev = torch.device("cuda:0")


def burst(fn, reps=10, rounds=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


for mb in (113, 453, 1359, 2718):
    n = mb * 1000 * 1000 // 4
    y = torch.empty((n,), dtype=torch.float32, device=dev)
    x = torch.rand((n,), dtype=torch.float32, device=dev)
    t_fill = burst(lambda: y.fill_(1.5))
    t_zero = burst(lambda: y.zero_())
    t_copy = burst(lambda: y.copy_(x))
    t_add = burst(lambda: torch.add(x, 1.0, out=y))
    print(f"{mb:5d} MB: fill {t_fill:.4f} ms = {n * 4 / t_fill / 1e9:5.2f} TB/s written | zero {t_zero:.4f} ms = {n * 4 / t_zero / 1e9:5.2f} | "
          f"copy {t_copy:.4f} ms = {n * 4 / t_copy / 1e9:5.2f} written ({2 * n * 4 / t_copy / 1e9:5.2f} moved) | add {t_add:.4f} ms = {2 * n * 4 / t_add / 1e9:5.2f} moved")

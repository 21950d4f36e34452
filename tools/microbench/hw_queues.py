#!/usr/bin/env python3
"""Which HIP streams of a process share a hardware queue: a 1-ms single-workgroup spin kernel on stream i and one on stream j --
2 ms if they serialise (same queue), 1 ms if they overlap.  python tools/microbench/hw_queues.py [nstreams]"""
import sys
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n)]
cycles = 2_000_000


def pair(a, b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    for s in (a, b):
        if s is not cur:
            s.wait_stream(cur)
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    for s in (a, b):
        if s is not cur:
            cur.wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for s in streams:
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()
one = pair(streams[1], streams[1])
print(f"two kernels on ONE stream: {one:.2f} ms (serial reference); index 0 = the default stream")
for i in range(len(streams)):
    row = []
    for j in range(len(streams)):
        row.append(" -- " if i == j else ("SAME" if pair(streams[i], streams[j]) > 0.75 * one else "  . "))
    print(f"{i:2d}: " + " ".join(row))

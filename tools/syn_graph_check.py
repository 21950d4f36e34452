"""The fused synthesis launch under HIP-graph capture / replay (its work queue is re-armed by a memset node): replays must give
the eager result.  python tools/syn_graph_check.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from shallow_ntc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
mk = lambda scale, *shape: torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).to(dev)
w1, b1 = mk(0.03, 13, 13, 24, 320), mk(0.1, 24)
beta = torch.from_numpy((1.0 + rng.random(12)).astype(np.float32)).to(dev)
gamma = torch.from_numpy((0.1 * np.eye(12) + 0.02 * rng.random((12, 12))).astype(np.float32)).to(dev)
syn = ops.SynPlan(w1, b1, 8, 12, True, 1, beta, gamma)
x = mk(1.0, 6, 32, 48, 320)
want = syn(x).clone()
torch.cuda.synchronize()
print("eager ok", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        syn(x)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = syn(x)
print("captured", flush=True)
for i in range(5):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, bool(torch.equal(out, want)), flush=True)

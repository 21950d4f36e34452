"""HIP-graph capture of the launch-bound decode sequence.

A decode pass is ~8 short kernels per batch; eager launches leave 5-10 us host gaps between them (and
~80 us between passes).  The whole sequence is captured once per shape through torch's graph API --
every kernel of libsntc_hip.so is launched on torch's current stream and nothing in the launch path
allocates or synchronises, so the capture is exact -- and replayed with one hipGraphLaunch."""
from __future__ import annotations

import torch

from . import ops


class DecodeGraph:
    """graph(z_hat, symbols) -> uint8 pixels [n, H, W, 3] (a static buffer, overwritten by the next call)."""

    def __init__(self, model, z_hat, symbols, image_hw, warmup=2):
        self.model, self.image_hw = model, tuple(image_hw)
        self.z_hat = z_hat.clone()
        self.symbols = symbols.clone()
        with torch.cuda.device(model.device):
            side = ops.side_streams(1, model.device)[0]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up off the capture: function attributes, allocator pools
                for _ in range(warmup):
                    model.decode(self.z_hat, self.symbols, self.image_hw, check=False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = model.decode(self.z_hat, self.symbols, self.image_hw, check=False)

    def __call__(self, z_hat=None, symbols=None):
        if z_hat is not None and z_hat.data_ptr() != self.z_hat.data_ptr():
            self.z_hat.copy_(z_hat)
        if symbols is not None and symbols.data_ptr() != self.symbols.data_ptr():
            self.symbols.copy_(symbols)
        self.graph.replay()
        return self.out



class DecodeSetGraph:
    """ONE graph for a set of batches of different image sizes: every batch's decode sequence is a branch of its own (captured on
    a stream forked from the capturing stream and joined back), so the branches run side by side as the eager two-stream decode
    does, without its host launches.  graph(codes=None) -> [uint8 pixels per batch] (static buffers)."""

    def __init__(self, model, codes, warmup=2):
        self.model = model
        self.codes = [(z.clone(), s.clone(), tuple(hw)) for z, s, hw, *_ in codes]
        with torch.cuda.device(model.device):
            self.streams = ops.side_streams(len(self.codes), model.device)
            for _ in range(warmup):
                self._launch()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._launch()

    def _launch(self):
        cur = torch.cuda.current_stream()
        outs = []
        for st, (z_hat, sym, hw) in zip(self.streams, self.codes):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self.model.decode(z_hat, sym, hw, check=False))
        for st in self.streams:
            cur.wait_stream(st)
        return outs

    def __call__(self, codes=None):
        if codes is not None:
            for (z0, s0, _hw), (z, s, *_r) in zip(self.codes, codes):
                if z.data_ptr() != z0.data_ptr():
                    z0.copy_(z)
                if s.data_ptr() != s0.data_ptr():
                    s0.copy_(s)
        self.graph.replay()
        return self.out

import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
graft.load_package()
from shallow_ntc_amd import ops
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
from shallow_ntc_amd.graphs import DecodeGraph
dev = torch.device("cuda:0")
model = Model(device=dev, **configs.two_layer_syn(rd_lambda=0.01))
g = torch.Generator(device=dev); g.manual_seed(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
if "nosk" in sys.argv:
    ops.set_stream_k(False)
for n, hw in ((6, (768, 512)), (18, (512, 768))):
    hp, wp = hw
    z_hat = torch.round(3.0 * torch.randn((n, hp // 64, wp // 64, 320), device=dev, generator=g)).contiguous()
    sym = torch.round(2.0 * torch.randn((n, hp // 16, wp // 16, 320), device=dev, generator=g)).to(torch.int32).contiguous()
    ref = model.decode(z_hat, sym, hw)
    torch.cuda.synchronize()
    print("eager", n, flush=True)
    if mode == "syn_only":
        hyper = model._hyper_synthesis(z_hat)
        y_hat = ops.dequant_scale_normal(sym, hyper)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = model._pixels(y_hat, hw)
        for i in range(3):
            gr.replay(); torch.cuda.synchronize()
            print("replay", i, bool(torch.equal(out, ref)), flush=True)
    else:
        dg = DecodeGraph(model, z_hat, sym, hw)
        print("captured", flush=True)
        for i in range(3):
            out = dg(); torch.cuda.synchronize()
            print("replay", i, bool(torch.equal(out, ref)), flush=True)

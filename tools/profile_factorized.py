import sys; sys.path.insert(0,'.')
import torch, numpy as np
import __graft_entry__ as g; g.load_package()
from shallow_ntc_amd import ops
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.factorized.models import Model
dev=torch.device('cuda:0')
m=Model(device=dev, **configs.bls2017())
for n,h,w in ((8,256,256),(18,512,768)):
    x=(torch.rand((n,h,w,3),device=dev)-.5).contiguous()
    with ops.autotune():
        m.end_to_end_frame_loss(x, training=False)
    m.end_to_end_frame_loss(x, training=False); torch.cuda.synchronize()
    acc={}
    for rep in range(5):
        ops.PROFILE=[]
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); m.end_to_end_frame_loss(x, training=False); e1.record(); torch.cuda.synchronize()
        for i,e in enumerate(ops.PROFILE):
            acc.setdefault(i,dict(e,samples=[]))["samples"].append(e["e0"].elapsed_time(e["e1"]))
        tot=e0.elapsed_time(e1); ops.PROFILE=None
    print(f"== bls2017 {n}x{h}x{w}: forward {tot:.3f} ms")
    for i,a in acc.items():
        ms=float(np.median(a["samples"])); print(f"{i:3d} {a['kind']:8s} k{a['k']} s{a['s']} {a['cin']:4d}->{a['cout']:4d} in {a['n']}x{a['h']}x{a['w']:<4d} {ms:8.4f} ms {a['flops']/ms/1e9:7.1f} TF {a['flops']/1e9:8.2f} GF")

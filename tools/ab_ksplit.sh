# A/B of the deterministic split-K threshold (conv_plan.hip::pick_ksplit, SNTC_KSPLIT_BPI_MAX) as LIBRARIES on one box:
#   lib/libsntc_hip.so (32: round 1's rule), lib/libsntc_k16.so, lib/libsntc_k8.so -- W1 decode (64 x 256 x 256) and one 256 x 256 image
# The variant libraries are not kept in the tree; build them here (CPU container) before the gpurun call:
#   cd shallow-ntc_amd/csrc && for v in 16 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSNTC_KSPLIT_BPI_MAX=$v \
#     -c conv_plan.hip -o /tmp/conv_plan_$v.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsntc_k$v.so \
#     $(ls build/*.o | grep -v conv_plan.o) /tmp/conv_plan_$v.o; done
# Result of round 6: profiles/r06_ab_ksplit.txt (the threshold stays at 32).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for i in 1 2; do
for lib in hip k16 k8; do
  export SNTC_LIB=$R/shallow-ntc_amd/lib/libsntc_$lib.so
  python bench.py --workload w1 --decode-only --no-cpu-baseline --steps 20 --warmup 5 --tuning-file '' > gpurun_out/ab_ks.json 2> gpurun_out/ab_ks.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_ks.json').read().strip().split('\n')[-1])
print('$lib: W1 decode', d['value'], 'Mpx/s', d['ms_per_step'], 'ms; region median', d['regions']['decode']['ms_per_step'])
PY
  python - <<PY
import sys; sys.path.insert(0,'.')
import torch, numpy as np
import __graft_entry__ as g; g.load_package()
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
dev=torch.device('cuda:0'); m=Model(device=dev, **configs.two_layer_syn())
for n,h,w in ((1,256,256),(8,256,256),(64,256,256),(1,512,768)):
    x=(torch.rand((n,h,w,3),device=dev)-.5).contiguous()
    z_hat,sym,_,_=m.encode(x)
    m.decode(z_hat,sym,(h,w)); torch.cuda.synchronize()
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): m.decode(z_hat,sym,(h,w),check=False)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m.encode(x,check=False)
    e1.record(); torch.cuda.synchronize()
    print(f"$lib: {n}x{h}x{w}: decode {np.median(ts):.4f} ms, encode {e0.elapsed_time(e1)/10:.4f} ms")
PY
done; done 2>&1 | grep -v amdgpu | tee gpurun_out/r06_ab_ksplit.txt

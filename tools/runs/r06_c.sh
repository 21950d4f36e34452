cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "hand_kats or factorized or entropy" 2>&1 | tail -5
python - <<'PY'
import sys; sys.path.insert(0,'.')
import torch, numpy as np
import __graft_entry__ as g; g.load_package()
from shallow_ntc_amd import ops
from shallow_ntc_amd.mshyper import configs
from shallow_ntc_amd.mshyper.models import Model
dev=torch.device('cuda:0'); m=Model(device=dev, **configs.two_layer_syn())
prior=m._get_prior()
for n in (18,1):
    z=torch.randn((n,8,12,320),device=dev)*3
    prior(z); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): prior(z)
    e1.record(); torch.cuda.synchronize()
    print(f"factorized n={n}: {e0.elapsed_time(e1)/50*1e3:.1f} us per call in a burst of 50")
PY

# configs[3] / configs[4] drivers on the real kernels: 1 rank vs 2 ranks (gloo, both ranks time-slicing this box's one GPU)
R=${GRAFT_REPO_ROOT:-.}; cd $R; O=gpurun_out
A="--images 4 --lambdas 0.01 0.04"
python3 tools/rd_sweep.py $A --out $O/rd_1rank.json > /dev/null 2> $O/rd_1rank.err
SNTC_SHARE_GPU=1 SNTC_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rd_sweep.py $A --out $O/rd_2rank.json > /dev/null 2> $O/rd_2rank.err
B="--images 4 --batch 2 --hw 256 256 --steps 200 --eval-every 100 --operating-point"
python3 tools/itinf_sweep.py $B --out $O/itinf_1rank.json > /dev/null 2> $O/itinf_1rank.err
SNTC_SHARE_GPU=1 SNTC_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/itinf_sweep.py $B --out $O/itinf_2rank.json > /dev/null 2> $O/itinf_2rank.err
python3 - <<'PY'
import json
for t in ("rd", "itinf"):
    a = json.load(open(f"gpurun_out/{t}_1rank.json")); b = json.load(open(f"gpurun_out/{t}_2rank.json"))
    na, nb = a.pop("n_gpus"), b.pop("n_gpus")
    print(t, "ranks", na, nb, "identical" if a == b else "DIFFERENT")
    print(json.dumps(a)[:600])
PY

# full GPU tier + default bench (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_full; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -15 $O/gpu_tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_full"
try:
    d=json.loads(open(O+"/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    for k,v in d["regions"].items():
        print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step","mpixels_per_s","speedup_over_fp32","roofline","pixels_differing_from_fp32","symbols_differing_from_fp32","psnr_only","psnr_only_serial")})
    print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_ms")})
    print("rccl", d["rccl"]["backend"], d["rccl"]["rccl_version"])
    print("cpu", d["cpu_baseline"])
except Exception as e:
    print("parse failed", e)
PY

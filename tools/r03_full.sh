# full GPU tier + default bench (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_full; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_full"
try:
    lines=open(O+"/bench.json").read().strip().splitlines()
    print("stdout lines:", len(lines))
    d=json.loads(lines[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    for k,v in d["regions"].items():
        print(k, v.get("ms_per_step"), v.get("mpixels_per_s"), (v.get("roofline") or {}).get("frac_of_fp32_mfma_peak"), v.get("speedup_over_fp32"))
    print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_ms")})
    print("rccl", d["rccl"]["backend"], d["rccl"]["rccl_version"])
except Exception as e:
    print("parse failed", e)
PY

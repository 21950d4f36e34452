#!/bin/bash
mkdir -p gpurun_out/r03_p12
python bench.py > gpurun_out/r03_p12/bench.json 2> gpurun_out/r03_p12/bench.err; tail -3 gpurun_out/r03_p12/bench.err
python - <<'PY'
import json
for f in ("bench",):
    d=json.load(open(f"gpurun_out/r03_p12/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("launch_schedule"))
    for k,v in d.get("regions",{}).items():
        print("  ", k, v.get("ms_per_step"), v.get("roofline",{}).get("frac_of_fp32_mfma_peak"), v.get("speedup_over_fp32"))
PY

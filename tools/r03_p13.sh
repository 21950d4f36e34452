#!/bin/bash
mkdir -p gpurun_out/r03_p13
python tools/profile_layers.py --reps 5 2>&1 | grep "^==\|conv total\|k5 s2" 
python bench.py --no-cpu-baseline --no-autotune > gpurun_out/r03_p13/b.json 2> gpurun_out/r03_p13/b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_p13/b.json"))
print("notune", d["value"], d["ms_per_step"])
for k,v in d.get("regions",{}).items():
    print("  ", k, v.get("ms_per_step"), v.get("roofline",{}).get("frac_of_fp32_mfma_peak"), v.get("speedup_over_fp32"))
PY

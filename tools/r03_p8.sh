#!/bin/bash
mkdir -p gpurun_out/r03_p8
timeout 900 python -m pytest tests/test_hip_bf16x3.py tests/test_hip_entry_points.py -x -q -m gpu > gpurun_out/r03_p8/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03_p8/t.log | head
python bench.py > gpurun_out/r03_p8/bench.json 2> gpurun_out/r03_p8/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_p8/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d.get("regions",{}).items():
    print(k, {kk: vv for kk,vv in v.items() if kk in ("ms","median_ms","speedup_over_fp32","mpx_s","roofline")})
PY

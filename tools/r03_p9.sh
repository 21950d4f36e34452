R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p9; mkdir -p $O
cd $R
python3 tools/profile_layers.py --reps 5 > $O/layer_table.txt 2>&1
python3 tools/profile_layers.py --reps 5 --batch 64 --hw 256 256 > $O/layer_table_w1.txt 2>&1
grep "==\|conv total" $O/layer_table.txt $O/layer_table_w1.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_p9/bench.json").read().splitlines()[0])
print("value", d["value"])
for k,v in d["regions"].items():
    print(k, v.get("ms_per_step"), v.get("mpixels_per_s"), (v.get("roofline") or {}).get("frac_of_fp32_mfma_peak"), v.get("speedup_over_fp32"))
PY

R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p14; mkdir -p $O
cd $R
python3 tools/profile_layers.py --reps 5 --decode-only 2>&1 | grep -v amdgpu
python3 tools/profile_layers.py --reps 5 --decode-only --batch 6 --hw 768 512 2>&1 | grep -v amdgpu
python3 tools/profile_layers.py --reps 5 2>&1 | grep "== encode\|conv total"
python bench.py --no-cpu-baseline --decode-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('value', d['value'], d['ms_per_step'], d['roofline']['frac'])"

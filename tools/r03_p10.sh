R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p10; mkdir -p $O
cd $R
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "row_packed" 2>&1 | tail -12
python -m pytest tests/test_hip_model.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -5
python3 tools/profile_layers.py --reps 5 > $O/layer_table.txt 2>&1
grep "==\|conv total\|   3-> 192" $O/layer_table.txt

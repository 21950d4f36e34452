R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p12; mkdir -p $O
cd $R
SNTC_FUSE2_T=8 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "fused_residual" 2>&1 | grep -E "passed|failed"
for t in 0 4 8 12 16 0 8; do
echo "naps $t"; SNTC_FUSE2_T=$t python3 tools/profile_layers.py --reps 5 2>&1 | grep "== encode\|  6 conv+1x1 k3 s1   96-> 192 in 18x256x384\| 19 conv+1x1\|conv total" | tail -4
done

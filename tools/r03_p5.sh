R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p5; mkdir -p $O
cd $R
python -m pytest tests/test_hip_bf16x3.py -m gpu -x -q 2>&1 | tail -5
run() { timeout 300 python3 $R/tools/one_layer.py "$@" --reps 6 >> $O/layers.txt 2>&1; }
: > $O/layers.txt
run --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --bf16x3
run --kind convT --k 5 --s 2 --cin 320 --cout 480 --n 18 --hw 16 24 --bf16x3
run --kind convT --k 13 --s 8 --cin 320 --cout 24 --n 18 --hw 32 48 --bf16x3
cat $O/layers.txt | grep -v "amdgpu.ids\|while staging"

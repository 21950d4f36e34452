R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p7; mkdir -p $O
cd $R
run() { timeout 300 python3 $R/tools/one_layer.py "$@" --reps 5 2>&1 | grep -v "amdgpu.ids\|while staging" >> $O/layers.txt; }
: > $O/layers.txt
run --kind conv --k 1 --s 1 --cin 4320 --cout 512 --n 16 --hw 64 64 --bf16x3
run --kind conv --k 3 --s 1 --cin 480 --cout 512 --n 16 --hw 64 64 --bf16x3
run --kind conv --k 1 --s 1 --cin 4320 --cout 512 --n 8 --hw 64 64 --bf16x3
cat $O/layers.txt
tools/microbench/gemm_ceiling 2>&1 | tail -4

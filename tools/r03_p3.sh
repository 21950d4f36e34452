R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p3; mkdir -p $O
$R/tools/microbench/gemm_ceiling > $O/ceiling.txt 2>&1
run() { python3 $R/tools/one_layer.py "$@" --reps 10 2>/dev/null >> $O/layers.txt; }
: > $O/layers.txt
run --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --variant 9
run --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --variant 9 --static
run --kind conv --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --variant 9
run --kind conv --k 1 --s 1 --cin 4320 --cout 640 --n 18 --hw 32 48 --variant 9
run --kind conv --k 1 --s 1 --cin 4320 --cout 640 --n 18 --hw 32 48 --variant 9 --static
run --kind conv --k 1 --s 1 --cin 4320 --cout 640 --n 16 --hw 32 64 --variant 9 --static
run --kind convT --k 13 --s 8 --cin 320 --cout 24 --n 18 --hw 32 48
run --kind conv --k 1 --s 1 --cin 1280 --cout 640 --n 18 --hw 32 48 --variant 9
run --kind conv --k 1 --s 1 --cin 1296 --cout 640 --n 18 --hw 32 48 --variant 9
cat $O/ceiling.txt $O/layers.txt

R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p13; mkdir -p $O
cd $R
run() { timeout 300 python3 $R/tools/one_layer.py "$@" --reps 8 2>&1 | grep -v "amdgpu.ids" >> $O/layers.txt; }
: > $O/layers.txt
run --kind convT --k 5 --s 2 --cin 320 --cout 480 --n 18 --hw 16 24 --ab 9,3,2,6
run --kind convT --k 5 --s 2 --cin 320 --cout 480 --n 6 --hw 24 16 --ab 9,3,8,2
run --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 6 --hw 48 32 --ab 9,8,4,5
run --kind convT --k 13 --s 8 --cin 320 --cout 24 --n 6 --hw 48 32 --ab 9,8,4,2
cat $O/layers.txt
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; grep -E "passed|failed|FAILED" $O/tests.txt | tail -3

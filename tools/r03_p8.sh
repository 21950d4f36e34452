R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p8; mkdir -p $O
cd $R
run() { timeout 300 python3 $R/tools/one_layer.py "$@" --reps 8 2>&1 | grep -v "amdgpu.ids" >> $O/layers.txt; }
: > $O/layers.txt
for st in "" "--static"; do
run --kind conv --k 3 --s 1 --cin 96 --cout 96 --n 18 --hw 256 384 $st
run --kind conv --k 1 --s 1 --cin 192 --cout 96 --n 18 --hw 256 384 $st
run --kind conv --k 1 --s 1 --cin 96 --cout 192 --n 18 --hw 256 384 --epi $st
run --kind conv --k 3 --s 1 --cin 96 --cout 96 --n 18 --hw 128 192 $st
run --kind conv --k 1 --s 1 --cin 192 --cout 96 --n 18 --hw 128 192 $st
run --kind conv --k 1 --s 1 --cin 192 --cout 192 --n 18 --hw 128 192 $st
run --kind conv --k 5 --s 2 --cin 192 --cout 192 --n 18 --hw 128 192 $st
run --kind conv --k 1 --s 1 --cin 320 --cout 160 --n 18 --hw 32 48 $st
run --kind conv --k 3 --s 1 --cin 160 --cout 160 --n 18 --hw 32 48 $st
done
cat $O/layers.txt

R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/ab_layers.txt
: > $O
run() { python3 $R/tools/one_layer.py "$@" --reps 12 2>/dev/null >> $O; echo >> $O; }
run --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --ab 2,4,5,9
run --kind convT --k 5 --s 2 --cin 320 --cout 480 --n 18 --hw 16 24 --ab 2,3,4,9
run --kind convT --k 13 --s 8 --cin 320 --cout 24 --n 18 --hw 32 48 --ab 2,4,9
run --kind conv --k 3 --s 1 --cin 320 --cout 320 --n 18 --hw 32 48 --ab 2,4,5,9

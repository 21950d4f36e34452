R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/ab_dma.txt
: > $O
run() { for d in 0 1 0 1; do python3 $R/tools/one_layer.py "$@" --dma $d --reps 10 2>/dev/null | sed "s/^/dma=$d /" >> $O; done; echo >> $O; }
run --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --ab 9,4
run --kind convT --k 5 --s 2 --cin 320 --cout 480 --n 18 --hw 16 24 --ab 9
run --kind convT --k 13 --s 8 --cin 320 --cout 24 --n 18 --hw 32 48 --ab 9
run --kind conv --k 3 --s 1 --cin 96 --cout 96 --n 18 --hw 256 384 --ab 3
run --kind conv --k 5 --s 2 --cin 192 --cout 192 --n 18 --hw 256 384 --ab 3
run --kind conv --k 1 --s 1 --cin 192 --cout 96 --n 18 --hw 256 384 --ab 3
run --kind conv --k 1 --s 1 --cin 96 --cout 192 --n 18 --hw 256 384 --epi --ab 3,8

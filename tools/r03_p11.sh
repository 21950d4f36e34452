#!/bin/bash
mkdir -p gpurun_out/r03_p11
export SNTC_LIB=$PWD/shallow-ntc_amd/lib/libsntc_diag.so
for spec in "convT 3 1 480 640 6 48 32" "convT 5 2 320 480 6 24 16" "convT 13 8 320 24 6 48 32" "convT 5 2 320 320 6 12 8" "convT 3 1 480 640 18 32 48" "convT 5 2 320 480 18 16 24" "convT 13 8 320 24 18 32 48" "convT 5 2 320 320 18 8 12"; do
  set -- $spec
  echo "== $spec"
  SNTC_SCHED_DBG=1 python tools/one_layer.py --kind $1 --k $2 --s $3 --cin $4 --cout $5 --n $6 --hw $7 $8 --reps 1 2>&1 | grep "sched" | sort -u
done > gpurun_out/r03_p11/sched.txt 2>&1
cat gpurun_out/r03_p11/sched.txt

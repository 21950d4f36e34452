#!/bin/bash
mkdir -p gpurun_out/r03_p7
timeout 600 python -m pytest tests/test_hip_bf16x3.py -x -q -m gpu > gpurun_out/r03_p7/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03_p7/t.log | head
for spec in "convT 3 1 480 640 18 32 48" "convT 5 2 320 480 18 16 24" "convT 3 1 480 640 6 48 32" "convT 5 2 320 480 6 24 16" ; do
  set -- $spec
  echo "== $spec"
  python tools/one_layer.py --kind $1 --k $2 --s $3 --cin $4 --cout $5 --n $6 --hw $7 $8 --bf16x3 --reps 12 2>&1 | grep -v "split while staging\|amdgpu.ids\|per-tap"
done > gpurun_out/r03_p7/layers.txt 2>&1
cat gpurun_out/r03_p7/layers.txt
export SNTC_LIB=$PWD/shallow-ntc_amd/lib/libsntc_diag.so
for dbg in 0 3 8 64 11 75; do
  echo "== SNTC_GG_DBG=$dbg"
  SNTC_GG_DBG=$dbg python tools/one_layer.py --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --bf16x3 --reps 6 2>&1 | grep "variant 12 stream-K patch"
done > gpurun_out/r03_p7/diag.txt 2>&1
cat gpurun_out/r03_p7/diag.txt

#!/bin/bash
mkdir -p gpurun_out/r03_p7
timeout 600 python -m pytest tests/test_hip_bf16x3.py -x -q -m gpu > gpurun_out/r03_p7/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03_p7/t.log | head
for spec in "convT 3 1 480 640 18 32 48" "convT 5 2 320 480 18 16 24" "convT 3 1 480 640 6 48 32" "convT 5 2 320 480 6 24 16"; do
  set -- $spec
  echo "== $spec"
  python tools/one_layer.py --kind $1 --k $2 --s $3 --cin $4 --cout $5 --n $6 --hw $7 $8 --bf16x3 --reps 12 2>&1 | grep -v "split while staging\|amdgpu.ids\|variant 11\|per-tap"
done > gpurun_out/r03_p7/layers.txt 2>&1
cat gpurun_out/r03_p7/layers.txt

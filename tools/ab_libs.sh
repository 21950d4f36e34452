#!/bin/bash
# A/B of the WORKING TREE's library against the library built from another commit's csrc/, on ONE box, interleaved.
# (Round 3: a branch added to the fp32 kernel's piece set-up moved its register allocation and cost the K loop 8 % -- and an A/B
# between the two code paths inside the NEW binary did not show it.  Compare binaries, not switches.)
#   here (CPU container):  tools/ab_libs.sh build <git-ref>      -> shallow-ntc_amd/lib/libsntc_ref.so from <git-ref>:shallow-ntc_amd/csrc
#   on the GPU box:        gpurun -- 'bash tools/ab_libs.sh run'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  ref=${2:?git ref}
  rm -rf $R/shallow-ntc_amd/csrc_ref && mkdir -p $R/shallow-ntc_amd/csrc_ref
  git -C $R archive $ref shallow-ntc_amd/csrc include | tar -x -C $R/shallow-ntc_amd/csrc_ref
  make -C $R/shallow-ntc_amd/csrc_ref/shallow-ntc_amd/csrc LIB=$R/shallow-ntc_amd/lib/libsntc_ref.so | tail -1
  exit 0
fi
O=$R/gpurun_out/ab_libs; mkdir -p $O
for i in 1 2; do
  for lib in hip ref; do
    export SNTC_LIB=$R/shallow-ntc_amd/lib/libsntc_$lib.so
    for spec in "convT 3 1 480 640 18 32 48" "conv 5 2 192 192 18 256 384" "conv 3 1 96 96 18 256 384" "conv 1 1 192 96 18 256 384"; do
      set -- $spec
      echo "$lib: $(python $R/tools/one_layer.py --kind $1 --k $2 --s $3 --cin $4 --cout $5 --n $6 --hw $7 $8 --reps 8 2>&1 | grep TFLOP | tail -1)"
    done
    python $R/bench.py --decode-only > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.load(open('$O/b.json')); print('$lib: bench decode', d['value'], 'Mpx/s', d['ms_per_step'], 'ms; region median', d['regions']['decode']['ms_per_step'])"
  done
done | tee $O/ab.txt

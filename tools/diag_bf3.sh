#!/bin/bash
# What does the pre-split bf16 x 3 loop (patch staging, 256 x 128, csrc/bf3_gemm.hip) wait on?  Builds the DIAG library
# (make DIAG=1: SNTC_DBG switches compiled in) next to the shipped one and times the 480 -> 640 layer with pieces of the K loop
# removed -- results are MEANINGLESS with a switch set, timing only:
#   1  no weight DMA after the first stages (the LDS keeps stale but real data)     2  no patch DMA after the first stages
#   4  no barriers      8  no fragment reads after the first two stages      64  no MFMAs
# Run through gpurun:  gpurun -- 'bash tools/diag_bf3.sh'
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/diag_bf3; mkdir -p $O
make -C $R/shallow-ntc_amd/csrc DIAG=1 BUILD=build_diag LIB=../lib/libsntc_diag.so > $O/build.log 2>&1
export SNTC_LIB=$R/shallow-ntc_amd/lib/libsntc_diag.so
for dbg in 0 3 4 7 8 11 15 64 75; do
  echo "== SNTC_GG_DBG=$dbg"
  SNTC_GG_DBG=$dbg python $R/tools/one_layer.py --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --bf16x3 --reps 6 2>&1 | grep "variant 12 stream-K patch"
done | tee $O/diag.txt

#!/bin/bash
# Where does the fused ResidualBlock tail's second phase spend its time?  DIAG build, timing only (results meaningless):
#   16 no epilogue stores   32 no residual loads   64 no MFMAs   128 no second contraction
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/diag_fused; mkdir -p $O
make -C $R/shallow-ntc_amd/csrc DIAG=1 BUILD=build_diag LIB=../lib/libsntc_diag.so > $O/build.log 2>&1
export SNTC_LIB=$R/shallow-ntc_amd/lib/libsntc_diag.so
for dbg in 0 16 32 48 128 144 176; do
  echo "== SNTC_GG_DBG=$dbg"
  SNTC_GG_DBG=$dbg python $R/tools/fused_tail.py --reps 8 2>&1 | grep "fused"
done | tee $O/diag.txt

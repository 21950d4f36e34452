#!/bin/bash
# What the fused synthesis launch waits on: a DIAG build (make -C shallow-ntc_amd/csrc DIAG=1 LIB=../lib/libsntc_hip_diag.so BUILD=build_diag)
# with parts of the kernel switched off by SNTC_SYN_DBG (results are meaningless with any bit set; timing only).
#   1 no epilogue, 2 no patch DMA, 4 no ring DMA, 8 no MFMAs, 16 no barriers
export SNTC_LIB=$PWD/shallow-ntc_amd/lib/libsntc_hip_diag.so
for dbg in 0 1 2 4 6 8 16 30 31; do
  echo "SNTC_SYN_DBG=$dbg"
  SNTC_SYN_DBG=$dbg timeout 120 python tools/syn_block.py 18 32 48 + 6 48 32 2>&1 | grep "fused"
done

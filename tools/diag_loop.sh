# what does the K loop wait on?  builds a -DSNTC_DIAG copy of the library in /tmp and times one layer with pieces of the loop removed
R=${GRAFT_REPO_ROOT:-.}
cd $R/shallow-ntc_amd/csrc && make -s clean >/dev/null 2>&1; make -s -j16 DIAG=1 >/dev/null 2>&1 || { echo "diag build failed"; exit 1; }
cd $R
O=$R/gpurun_out/diag_loop.txt; : > $O
for layer in "--kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48" "--kind conv --k 5 --s 2 --cin 192 --cout 192 --n 18 --hw 256 384"; do
for dbg in 0 1 2 4 8 3 7 15; do
  echo "SNTC_GG_DBG=$dbg" >> $O
  SNTC_GG_DBG=$dbg python3 tools/one_layer.py $layer --reps 8 2>/dev/null >> $O
done; done
cd $R/shallow-ntc_amd/csrc && make -s clean >/dev/null 2>&1

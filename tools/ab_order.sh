#!/bin/bash
# fp32 stream-K unit order on the decode layers: strip-major (1) against column-major (0): time and HBM-side bytes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_order; rm -rf $O; mkdir -p $O
make -C $R/shallow-ntc_amd/csrc DIAG=1 BUILD=build_diag LIB=../lib/libsntc_diag.so > $O/build.log 2>&1      # SNTC_SK_ORDER is read by DIAG builds only
export SNTC_LIB=$R/shallow-ntc_amd/lib/libsntc_diag.so
for spec in "convT 3 1 480 640 18 32 48" "convT 5 2 320 480 18 16 24" "convT 3 1 480 640 6 48 32"; do
  set -- $spec
  for ord in 1 0; do
    export SNTC_SK_ORDER=$ord
    L="python3 $R/tools/one_layer.py --kind $1 --k $2 --s $3 --cin $4 --cout $5 --n $6 --hw $7 $8 --reps 8"
    echo "== $spec order $ord: $($L 2>&1 | grep TFLOP | tail -1)"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$ord -- $L > /dev/null 2>&1
    python3 - $O/f_$ord <<'PY'
import csv, glob, sys
v=[float(r["Counter_Value"]) for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gg_kernel" in r["Kernel_Name"]]
print("   FETCH_SIZE per launch: %.1f MB x2 = %.1f MB HBM-side reads" % (sum(v)/len(v)*1024/1e6, 2*sum(v)/len(v)*1024/1e6))
PY
    rm -rf $O/f_$ord
  done
done

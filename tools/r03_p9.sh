cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_p9; rm -rf $O; mkdir -p $O
cd $R/tools/microbench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mstats -- ./gemm_ceiling > $O/mb.txt 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/msq -- ./gemm_ceiling > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03_p9"
tr=collections.defaultdict(list)
for f in glob.glob(O+"/mstats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tr[r["Kernel_Name"]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/msq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "bf3" not in k: continue
    us=max(tr[k]) if tr[k] else 0
    gui=max(v["GRBM_GUI_ACTIVE"]); mf=max(v["SQ_VALU_MFMA_BUSY_CYCLES"])
    print(k[:80], "us %.0f clk %.2f GHz mfma_busy/(us*2.4*1024) %.3f" % (us, gui/8/us/1e3 if us else 0, mf/(us*2.4*1024*1e3) if us else 0))
PY
python tools/one_layer.py --kind convT --k 3 --s 1 --cin 1920 --cout 640 --n 18 --hw 32 48 --bf16x3 --reps 6 2>&1 | grep "variant 12"

R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r03_p11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
L="python3 $R/tools/profile_layers.py --reps 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -- $L > /dev/null 2>&1
SNTC_NO_ROWPACK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/norp -- $L > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03_p11"
for d in ("rp","norp"):
    tot=0; rows=[]
    for f in glob.glob(f"{O}/{d}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tot+=float(r["TotalDurationNs"]); rows.append((float(r["TotalDurationNs"]), int(r["Calls"]), r["Name"][:90]))
    print(d, "total GPU ms %.2f" % (tot/1e6))
    for t,c,n in sorted(rows, reverse=True)[:12]: print("   %.3f ms %5d %s" % (t/1e6,c,n))
PY
cd $R
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; grep -E "passed|failed|FAILED|Error" $O/tests.txt | tail -5

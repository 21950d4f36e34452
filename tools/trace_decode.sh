#!/bin/bash
# kernel + memory-copy trace of a few serial decode steps: which small launches sit between the convolutions?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_decode; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -- python3 $R/bench.py --decode-only --streams 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/err.txt
python3 - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/trace_decode"
ev=[]
for f in glob.glob(O+"/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(O+"/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size",""))))
ev.sort()
# last 140 events = the last timed steps
t0=ev[-140][0]
prev=None
for s,e,n in ev[-140:]:
    gap = (s-prev)/1e3 if prev else 0
    print(f"{(s-t0)/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  gap {gap:7.1f}  {n}")
    prev=e
PY

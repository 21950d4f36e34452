#!/bin/bash
# kernel trace (start / end timestamps per stream) of the default two-stream decode: which launches really run side by side?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_two; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --decode-only --steps 4 --warmup 2 > $O/bench.json 2> $O/err.txt
python3 - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/trace_two"
ev=[]
for f in glob.glob(O+"/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64], r.get("Queue_Id", r.get("Stream_Id", "?"))))
ev.sort()
# find the 4 timed steps of the headline: take events in the window of the first 'value' timed loop: print 60 events after the 3rd occurrence of the COLM kernel
idx=[i for i,e in enumerate(ev) if "true>(sntc::GGArgs)" in e[2] and "2, 2, 2, 2" in e[2]]
start=max(0, idx[3]-12) if len(idx)>3 else 0
t0=ev[start][0]
for s,e,n,q in ev[start:start+44]:
    print(f"{(s-t0)/1e3:9.1f} -> {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  q{q}  {n}")
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_p6; mkdir -p $O
L="python3 $R/tools/one_layer.py --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --bf16x3 --reps 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $L > $O/a.txt 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $L > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $L > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $L > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03_p6"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sntc" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
st={}
for f in glob.glob(O+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        st[r["Name"]]=(float(r["AverageNs"]), int(r["Calls"]))
for k,v in agg.items():
    m={c: sum(x)/len(x) for c,x in v.items()}
    a=st.get(k,(0,0))
    if a[0] < 100000: continue
    print(k[:70], "avg_us %.1f calls %d" % (a[0]/1e3, a[1]))
    print("   mfma_busy %.3f clk %.2f wait_any %.2f wait_inst %.2f active %.2f fetchMB %.1f writeMB %.1f ldsconf %.0f" % (
        m.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(a[0]*2.4*1024) if a[0] else 0, m.get("GRBM_GUI_ACTIVE",0)/8/a[0] if a[0] else 0,
        m.get("SQ_WAIT_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1), m.get("SQ_WAIT_INST_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1),
        m.get("SQ_ACTIVE_INST_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1), 2*m.get("FETCH_SIZE",0)*1024/1e6, m.get("WRITE_SIZE",0)*1024/1e6, m.get("SQ_LDS_BANK_CONFLICT",0)))
PY

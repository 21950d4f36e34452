# what the waves of the 1x1 + skip layer wait for: SQ counters (two passes) + memory-side counters, one layer in a loop
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_1x1; mkdir -p $O
L="python3 $R/tools/one_layer.py --kind conv --k 1 --s 1 --cin 96 --cout 192 --n 18 --hw 256 384 --reps 4 --epi"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/a -- $L > $O/a.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU --output-format csv -d $O/b -- $L > $O/b.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $O/c -- $L > $O/c.txt 2>&1
# (a fourth pass with TCP_TCC_READ_REQ_sum / TCC_HIT_sum / TCC_MISS_sum aborted inside rocprofv3 and hung the call: not collected)
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_1x1"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gg_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(f"{k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
tail -2 $O/a.txt | cut -c1-200

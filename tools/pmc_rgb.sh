# counters of the RGB first-layer kernel on the 18 x 512 x 768 shape (kernel stats + counter passes); gpurun -- 'bash tools/pmc_rgb.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rgb; rm -rf $O; mkdir -p $O
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
CMD="python3 $R/tools/rgb_conv_block.py --shapes 18x512x768 --rounds 2 --reps 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.txt 2> $O/stats.err
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq -- $CMD > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $CMD > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $CMD > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $O/pmc_inst -- $CMD > /dev/null 2> $O/pmc_inst.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 TCP_TCC_WRITE_REQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $O/pmc_wr -- $CMD > /dev/null 2> $O/pmc_wr.err
cd $R
python3 tools/summarize_pmc.py tmp_rgb_pmc $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_inst $O/pmc_wr > $O/summary.txt 2>&1
cp profiles/tmp_rgb_pmc.json $O/rgb_pmc_summary.json; rm -f profiles/tmp_rgb_pmc.json
tail -3 $O/pmc_wr.err
python3 - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_rgb/rgb_pmc_summary.json"))
for k,v in d.items():
    if "rgb_conv" in k or "gg_kernel" in k or "pad" in k: print(k[:90], json.dumps(v,indent=1))
PY

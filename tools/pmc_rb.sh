# counters of the whole-ResidualBlock kernel on the 18 x 256 x 384 shape (kernel stats + three --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rb; rm -rf $O; mkdir -p $O
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/rb_block.py 18 256 384 > $O/stats.txt 2> $O/stats.err
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq -- python3 $R/tools/rb_block.py 18 256 384 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/rb_block.py 18 256 384 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/rb_block.py 18 256 384 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $O/pmc_inst -- python3 $R/tools/rb_block.py 18 256 384 > /dev/null 2> $O/pmc_inst.err
cd $R
python3 tools/summarize_pmc.py tmp_rb_pmc $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_inst > $O/summary.txt 2>&1
cp profiles/tmp_rb_pmc.json $O/rb_pmc_summary.json; rm -f profiles/tmp_rb_pmc.json
cat $O/stats.txt | tail -2

# stream kernels: times + the counters that say what bounds scale_normal (VALU issue, not HBM)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_stream; rm -rf $O; mkdir -p $O
python3 $R/tools/stream_kernels.py > $O/stream_kernels.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/stream_kernels.py > /dev/null 2> $O/stats.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/tools/stream_kernels.py > /dev/null 2> $O/pmc.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/stream_kernels.py > /dev/null 2>> $O/pmc.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/stream_kernels.py > /dev/null 2>> $O/pmc.err
cd $R
python3 tools/summarize_pmc.py tmp_stream_pmc $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/pmc_sq $O/pmc_fetch $O/pmc_write > $O/summary.txt 2>&1
cp profiles/tmp_stream_pmc.json $O/stream_pmc_summary.json; rm -f profiles/tmp_stream_pmc.json
cat $O/stream_kernels.txt | tail -12

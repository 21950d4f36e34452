# HBM-side bytes of the 480 -> 640 hyper-synthesis launch under the two stream-K unit orders (tools/ab_order.py launches both
# instances in one process; their kernel names differ in the last template argument).  bash tools/pmc_order.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_order; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/ab_order.py --reps 6 > $O/ab_order.txt 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/ab_order.py --reps 6 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/ab_order.py --reps 6 > /dev/null 2> $O/write.err
cd $R
python3 tools/summarize_pmc.py r04_order_pmc_summary $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/fetch $O/write > $O/summary.txt 2>&1
cat $O/summary.txt | tail -5

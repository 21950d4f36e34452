# round-3 probe 1: ceiling microbench, encode-side counters of the current build, bf16x3 layer A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_probe1; mkdir -p $O
$R/tools/microbench/gemm_ceiling > $O/ceiling.txt 2>&1
L="python3 $R/tools/profile_layers.py --reps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_stats -- $L > $O/layer_table.txt 2> $O/enc_stats.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/enc_pmc_sq -- $L > /dev/null 2> $O/enc_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/enc_pmc_fetch -- $L > /dev/null 2> $O/enc_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/enc_pmc_write -- $L > /dev/null 2> $O/enc_pmc_write.err
cd $R
python3 tools/one_layer.py --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --reps 10 --bf16x3 > $O/bf3_layers.txt 2>&1
python3 tools/one_layer.py --kind conv --k 5 --s 2 --cin 192 --cout 192 --n 18 --hw 256 384 --reps 6 --bf16x3 >> $O/bf3_layers.txt 2>&1
python3 tools/one_layer.py --kind conv --k 3 --s 1 --cin 96 --cout 96 --n 18 --hw 256 384 --reps 6 --bf16x3 >> $O/bf3_layers.txt 2>&1
python3 tools/one_layer.py --kind conv --k 5 --s 2 --cin 192 --cout 192 --n 18 --hw 256 384 --reps 6 --ab 3,6,9 >> $O/bf3_layers.txt 2>&1
find $O -name "*.csv" | head -30
cat $O/ceiling.txt

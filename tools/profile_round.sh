# rocprofv3 evidence for profiles/ (round 3): kernel stats + PMC passes of the decode-only bench AND of the encode-side layer
# table, the layer tables (Kodak batch and W1), the loop-ceiling microbench, the default bench line.  Program directly after `--`.
# NOTE: gpurun MERGES what this writes into the local gpurun_out/prof_r03 -- delete that directory locally before a re-run, or
# stale *_kernel_stats.csv / *_counter_collection.csv of the previous run are averaged into tools/summarize_pmc.py's output.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; rm -rf $O; mkdir -p $O
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
# ---- decode (the headline region): one stream, so that a kernel's duration is its own
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec_stats -- python3 $R/bench.py --decode-only --streams 1 --steps 20 --warmup 3 > $O/decode_only_bench.json 2> $O/dec_stats.err
rocprofv3 --pmc $SQ --output-format csv -d $O/dec_pmc_sq -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 > /dev/null 2> $O/dec_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dec_pmc_fetch -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 > /dev/null 2> $O/dec_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dec_pmc_write -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 > /dev/null 2> $O/dec_pmc_write.err
# ---- encode (ELIC analysis + hyper transforms): the layer table script launches every layer of encode and decode
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_stats -- python3 $R/tools/profile_layers.py --reps 5 > $O/layer_table.txt 2> $O/enc_stats.err
rocprofv3 --pmc $SQ --output-format csv -d $O/enc_pmc_sq -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/enc_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/enc_pmc_fetch -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/enc_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/enc_pmc_write -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/enc_pmc_write.err
# ---- un-profiled: layer tables, microbench, default bench
cd $R
python3 tools/profile_layers.py --reps 5 > $O/layer_table_unprofiled.txt 2>&1
python3 tools/profile_layers.py --reps 5 --autotune > $O/layer_table_tuned.txt 2>&1
python3 tools/profile_layers.py --reps 5 --batch 64 --hw 256 256 --autotune > $O/layer_table_w1.txt 2>&1
( cd tools/microbench && echo "# tools/microbench/gemm_ceiling (raw float bits as operands)" && ./gemm_ceiling && echo && echo "# GEMM_CEILING_S3=1: genuine S3 operands (hi / mid / lo terms of N(0,1) values) for the bf16x3 loops" && GEMM_CEILING_S3=1 ./gemm_ceiling | grep -i bf16x3 ) > $O/gemm_ceiling.txt 2>&1
python3 tools/one_layer.py --kind convT --k 3 --s 1 --cin 480 --cout 640 --n 18 --hw 32 48 --bf16x3 --reps 12 > $O/bf16x3_hs3.txt 2>&1
python3 tools/one_layer.py --kind convT --k 5 --s 2 --cin 320 --cout 480 --n 18 --hw 16 24 --bf16x3 --reps 12 > $O/bf16x3_hs2.txt 2>&1
python3 tools/one_layer.py --kind convT --k 13 --s 8 --cin 320 --cout 24 --n 18 --hw 32 48 --bf16x3 --reps 12 > $O/bf16x3_syn.txt 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head; ls $O

# rocprofv3 evidence for profiles/: kernel stats + PMC passes of the decode-only bench, kernel stats + SQ pass of the layer table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02; mkdir -p $O
B="python3 $R/bench.py --decode-only --streams 1 --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/decode_only_bench.json 2> $O/stats.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/layers_stats -- python3 $R/tools/profile_layers.py --reps 5 > $O/layer_table.txt 2> $O/layers.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/layers_pmc -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/layers_pmc.err
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
ls -R $O | head -50

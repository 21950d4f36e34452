# rocprofv3 evidence for profiles/ (round 6): kernel stats + PMC passes of the decode-only bench AND of the encode-side layer
# table, the layer tables (Kodak batch and W1), the whole-ResidualBlock kernel alone, the stream kernels, the default bench line.
# Program directly after `--`.  bash tools/profile_round.sh [round tag, default r04]
# NOTE: gpurun MERGES what this writes into the local gpurun_out/prof_<tag> -- delete that directory locally before a re-run, or
# stale *_kernel_stats.csv / *_counter_collection.csv of the previous run are averaged into tools/summarize_pmc.py's output.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
# ---- decode (the headline region): one stream, so that a kernel's duration is its own; the launches are the ones bench.py's
# two-stream step settles on (ops.tune_step) -- measured ONCE here, unprofiled, and applied from the file in the profiled runs
python3 $R/bench.py --decode-only --no-cpu-baseline --steps 5 --warmup 2 --retune --write-tuning $O/decode_tuning.json > $O/decode_tuning_run.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec_stats -- python3 $R/bench.py --decode-only --streams 1 --steps 20 --warmup 3 --tuning-file $O/decode_tuning.json > $O/decode_only_bench.json 2> $O/dec_stats.err
rocprofv3 --pmc $SQ --output-format csv -d $O/dec_pmc_sq -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 --tuning-file $O/decode_tuning.json > /dev/null 2> $O/dec_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dec_pmc_fetch -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 --tuning-file $O/decode_tuning.json > /dev/null 2> $O/dec_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dec_pmc_write -- python3 $R/bench.py --decode-only --streams 1 --steps 5 --warmup 2 --tuning-file $O/decode_tuning.json > /dev/null 2> $O/dec_pmc_write.err
# ---- encode (ELIC analysis + hyper transforms): the layer table script launches every layer of encode and decode
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_stats -- python3 $R/tools/profile_layers.py --reps 5 > $O/layer_table_profiled.txt 2> $O/enc_stats.err
rocprofv3 --pmc $SQ --output-format csv -d $O/enc_pmc_sq -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/enc_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/enc_pmc_fetch -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/enc_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/enc_pmc_write -- python3 $R/tools/profile_layers.py --reps 3 > /dev/null 2> $O/enc_pmc_write.err
# ---- SGA step and the bitstream: kernel stats of the commands behind bench.py's sga_step / compress / decompress regions
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sga_stats -- python3 $R/tools/profile_sga.py --batch 5 --hw 1200 1200 --config two_layer_syn2 --steps 20 > $O/sga_step.txt 2> $O/sga_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bits_stats -- python3 $R/tools/profile_bitstream.py --batch 18 --reps 5 > $O/bitstream.txt 2> $O/bits_stats.err
# ---- un-profiled: layer tables, the ResidualBlock kernel against the layers it replaces, stream kernels, default bench
cd $R
python3 tools/profile_layers.py --reps 5 > $O/layer_table.txt 2>&1
python3 tools/profile_layers.py --reps 5 --autotune > $O/layer_table_tuned.txt 2>&1
python3 tools/profile_layers.py --reps 5 --batch 64 --hw 256 256 --autotune > $O/layer_table_w1.txt 2>&1
python3 tools/rb_block.py > $O/resblock_vs_layers.txt 2>&1
python3 tools/rgb_conv_block.py > $O/rgb_conv_vs_rowpacked.txt 2>&1
python3 tools/encoder_segments.py > $O/encoder_segments.txt 2>&1
SNTC_NO_BRANCH_STREAMS=1 python3 tools/encoder_segments.py --batches 18x512x768 > $O/encoder_segments_no_branch_streams.txt 2>&1
python3 tools/ab_encode.py > $O/ab_encode_rgb.txt 2>&1
python3 tools/ab_encode.py --one-stream >> $O/ab_encode_rgb.txt 2>&1
python3 tools/microbench/write_bw.py > $O/write_bw.txt 2>&1
python3 tools/syn_block.py > $O/synthesis_vs_layers.txt 2>&1
python3 tools/syn_block.py --ch 24 --res 0 5 76 76 >> $O/synthesis_vs_layers.txt 2>&1
python3 bench.py --decode-only --steps 50 --warmup 10 --set-decode > $O/decode_only_set_decode.json 2> /dev/null
python3 bench.py --decode-only --steps 50 --warmup 10 --streams 1 > $O/decode_only_one_stream.json 2> /dev/null
python3 tools/stream_kernels.py > $O/stream_kernels.txt 2>&1
python3 tools/time_evaluate.py > $O/evaluate_b1.txt 2>&1
python3 tools/profile_layers.py --reps 5 --precision bf16x3 > $O/layer_table_bf16x3.txt 2>&1
python3 tools/rans_steps.py > $O/rans_steps.txt 2>&1
python3 tools/rans_steps.py --no-lut >> $O/rans_steps.txt 2>&1
python3 tools/decompress_phases.py > $O/decompress_phases.txt 2>&1
python3 tools/decompress_phases.py --one-by-one >> $O/decompress_phases.txt 2>&1
python3 tools/ab_order.py > $O/ab_order.txt 2>&1
python3 tools/summarize_pmc.py ${TAG}_pmc_summary $(find $O/dec_stats -name "*kernel_stats.csv" | head -1) $O/dec_pmc_sq $O/dec_pmc_fetch $O/dec_pmc_write > $O/summary_dec.txt 2>&1
python3 tools/summarize_pmc.py ${TAG}_encode_pmc_summary $(find $O/enc_stats -name "*kernel_stats.csv" | head -1) $O/enc_pmc_sq $O/enc_pmc_fetch $O/enc_pmc_write > $O/summary_enc.txt 2>&1
# (the summaries above carry the hashes of the kernel sources they were measured on: the bench line below reads them, traffic_stale false)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cp profiles/${TAG}_pmc_summary.json profiles/${TAG}_encode_pmc_summary.json $O/
for d in dec enc sga bits; do cp $(find $O/${d}_stats -name "*kernel_stats.csv" | head -1) $O/${d}_kernel_stats.csv; done
ls $O

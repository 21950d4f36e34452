cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_gputest_b.txt; cat gpurun_out/r06_gputest_b.txt
python bench.py --steps 20 --warmup 5 --retune > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err; tail -c 600 gpurun_out/r06_bench_b.json; tail -3 gpurun_out/r06_bench_b.err
ls -la gpurun_out/tuning_gfx950.json

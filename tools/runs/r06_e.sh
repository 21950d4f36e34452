cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_gputest_e.txt; cat gpurun_out/r06_gputest_e.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4

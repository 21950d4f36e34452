set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "rgb or row_packed" 2>&1 | tail -15
python tools/rgb_conv_block.py > gpurun_out/r06_rgb_conv.txt 2>&1; cat gpurun_out/r06_rgb_conv.txt

#!/bin/bash
mkdir -p gpurun_out/r03_p10
( cd tools/microbench && echo "# tools/microbench/gemm_ceiling (raw float bits as operands)" && ./gemm_ceiling && echo && echo "# GEMM_CEILING_S3=1: genuine S3 operands (hi / mid / lo terms of N(0,1) values) for the bf16x3 loops" && GEMM_CEILING_S3=1 ./gemm_ceiling | grep -i bf16x3 ) > gpurun_out/r03_p10/gemm_ceiling.txt 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_p10/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03_p10/t.log | head

#!/usr/bin/env bash
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1" > gpurun_out/r03_conv_tests.log 2>&1 || { tail -30 gpurun_out/r03_conv_tests.log; exit 1; }
tail -2 gpurun_out/r03_conv_tests.log
for ns in 2 4; do
echo "== DSKD_GEMM_NS=$ns"
DSKD_GEMM_NS=$ns timeout -k 10 600 python scratch/r03_conv1x1.py > gpurun_out/r03_conv1x1_ns$ns.txt 2>&1 || { tail -30 gpurun_out/r03_conv1x1_ns$ns.txt; exit 1; }
cat gpurun_out/r03_conv1x1_ns$ns.txt
done

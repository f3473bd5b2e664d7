#!/usr/bin/env bash
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
for v in "DSKD_GEMM_BM=64" "DSKD_GEMM_BM=128" "DSKD_X=auto"; do
tag=$(echo $v | tr ' =' '__')
env $v timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1" > gpurun_out/r03_conv_tests_$tag.log 2>&1 || { tail -30 gpurun_out/r03_conv_tests_$tag.log; exit 1; }
tail -1 gpurun_out/r03_conv_tests_$tag.log
echo "== $v"
env $v timeout -k 10 600 python scratch/r03_conv1x1.py > gpurun_out/r03_conv1x1_$tag.txt 2>&1 || { tail -30 gpurun_out/r03_conv1x1_$tag.txt; exit 1; }
grep -v amdgpu gpurun_out/r03_conv1x1_$tag.txt
done

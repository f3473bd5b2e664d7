set -euo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or conv1x1 or gemm_tn or ffn or lin256 or tall_linear" > gpurun_out/r03_conv3_tests.log 2>&1 || { tail -40 gpurun_out/r03_conv3_tests.log; exit 1; }
tail -1 gpurun_out/r03_conv3_tests.log

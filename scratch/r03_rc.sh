#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_tn or bottleneck or conv1x1" 2>&1 | tail -3 || exit 1
python scratch/r03_tn_bench.py 2>&1 | grep "us " | head -18
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graphed or full_step" 2>&1 | tail -3 || exit 1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_rc.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide reduce_cvt', d['value'], d['ms_per_step'])"; done

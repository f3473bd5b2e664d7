#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_tn or bottleneck or conv1x1 or ffn_fused_vs" 2>&1 | tail -5 || exit 1
python scratch/r03_tn_bench.py 2>&1 | grep "us " | head -8
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_dw.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planes', d['value'], d['ms_per_step'])"
DSKD_DW_ATOMIC=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_dw.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('atomic', d['value'], d['ms_per_step'])"
done

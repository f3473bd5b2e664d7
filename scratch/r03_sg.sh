#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graphed or full_step or golden or variants or decode" 2>&1 | tail -5 || exit 1
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_sg_on.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('short gemm own', d['value'], d['ms_per_step'])"
DSKD_SHORT_GEMM_LIB=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_sg_off.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('short gemm lib', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_sg_on.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('short gemm own', d['value'], d['ms_per_step'])"

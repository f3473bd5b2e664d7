#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graphed or full_step or golden" 2>&1 | tail -3 || exit 1
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_dx.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('own dX', d['value'], d['ms_per_step'])"
DSKD_SHORT_DX_LIB=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_dx.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib dX', d['value'], d['ms_per_step'])"
done

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "prepack" 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_pp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prepack', d['value'], d['ms_per_step'])"
DSKD_NO_PREPACK=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_pp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('packs on the spot', d['value'], d['ms_per_step'])"
done

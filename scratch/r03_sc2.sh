#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_sc.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent', d['value'], d['ms_per_step'])"
DSKD_ACC_FRESH=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_sc.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fresh', d['value'], d['ms_per_step'])"
done

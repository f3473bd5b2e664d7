#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "gfl" 2>&1 | tail -5 || exit 1
for v in 0 1; do
  if [ $v = 1 ]; then export DSKD_NECK3X3_OFF=1; fi
  python bench.py --steps 10 --backbone gfl_r50 --no-cpu-baseline --no-mfma-probe 2>gpurun_out/r03_gfl3_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gfl off=$v', d['value'], d['ms_per_step'])"
done

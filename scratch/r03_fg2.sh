#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fgkd or corr" 2>&1 | tail -3
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --probe-steps 0 > gpurun_out/r03_b.json 2> gpurun_out/r03_b.err; python -c "
import json; d=json.load(open('gpurun_out/r03_b.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d['mfma'])[:1200])"

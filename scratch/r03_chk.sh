#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "msda" > gpurun_out/r03_mm_tests.log 2>&1; tail -2 gpurun_out/r03_mm_tests.log
for rep in 1 2; do
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_step_bench_$rep.json 2> gpurun_out/r03_step_bench_$rep.err || { tail -20 gpurun_out/r03_step_bench_$rep.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_step_bench_$rep.json')); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
done

#!/usr/bin/env bash
# model-level GPU tests + two bench runs
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "${1:-full_step or graphed or gfl}" > gpurun_out/r03_step_tests.log 2>&1 || { tail -40 gpurun_out/r03_step_tests.log; exit 1; }
tail -1 gpurun_out/r03_step_tests.log
for rep in 1 2; do
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_step_bench_$rep.json 2> gpurun_out/r03_step_bench_$rep.err || { tail -20 gpurun_out/r03_step_bench_$rep.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_step_bench_$rep.json')); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
done

#!/usr/bin/env bash
# the whole -m gpu suite (one process), then two bench runs
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gpu_tests.log
tail -4 gpurun_out/r03_gpu_tests.log
for rep in 1 2; do
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_step_bench_$rep.json 2> gpurun_out/r03_step_bench_$rep.err || { tail -20 gpurun_out/r03_step_bench_$rep.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_step_bench_$rep.json')); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
done

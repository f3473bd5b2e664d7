#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_step_bench_$rep.json 2> gpurun_out/r03_step_bench_$rep.err || { tail -20 gpurun_out/r03_step_bench_$rep.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_step_bench_$rep.json')); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
bash scratch/r02_breakdown.sh > /dev/null 2>&1; cp gpurun_out/bd_step_breakdown.txt gpurun_out/r03_step_breakdown_mid3.txt; head -12 gpurun_out/bd_step_breakdown.txt; grep -c . gpurun_out/bd_step_breakdown.txt; grep "dense_loss\|n kernels" gpurun_out/bd_step_breakdown.txt

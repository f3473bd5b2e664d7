#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "fused_prologue or teacher or full_step or bf16_tall" 2>&1 | tail -4
for rep in 1 2; do
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_step_bench_$rep.json 2> gpurun_out/r03_step_bench_$rep.err || { tail -20 gpurun_out/r03_step_bench_$rep.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_step_bench_$rep.json')); print(d['ms_per_step'], d['value'], d['config']['final_loss']); k=d['roofline']['kernels']; print({n: v['avg_us'] for n, v in k.items()})"
done

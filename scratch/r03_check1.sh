#!/usr/bin/env bash
# round 3, first GPU check: the new / changed -m gpu tests, then the default bench line
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu \
  -k "full_step or logit_and_local or ffn_fused_vs_float or lin256_vs_float or graphed_student" > gpurun_out/r03_check1_tests.log 2>&1
tail -3 gpurun_out/r03_check1_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err
cat gpurun_out/r03_bench0.json | head -c 600

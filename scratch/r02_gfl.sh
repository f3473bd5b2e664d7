#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "gfl_distillation" > gpurun_out/r02_gfl_test.log 2>&1; rc=$?
grep -E "^E |passed|failed" gpurun_out/r02_gfl_test.log | cut -c1-300 | tail -10
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --backbone gfl_r50 --steps 10 --no-cpu-baseline --no-mfma-probe > gpurun_out/r02_bench_gfl.json 2> gpurun_out/r02_bench_gfl.err; rc=$?
tail -3 gpurun_out/r02_bench_gfl.err | cut -c1-300; cut -c1-900 gpurun_out/r02_bench_gfl.json
exit $rc

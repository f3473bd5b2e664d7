#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gpu_tests.log
tail -4 gpurun_out/r03_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1; tail -2 gpurun_out/r03_smoke.log

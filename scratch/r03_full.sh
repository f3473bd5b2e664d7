#!/usr/bin/env bash
# the whole -m gpu suite (one process), then the kernel breakdown of the default bench
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gpu_tests.log
tail -4 gpurun_out/r03_gpu_tests.log
bash scratch/r02_breakdown.sh > /dev/null 2>&1; cp gpurun_out/bd_step_breakdown.txt gpurun_out/r03_step_breakdown_mid2.txt; head -64 gpurun_out/bd_step_breakdown.txt

#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "msda" > gpurun_out/r02_pull_tests.log 2>&1; rc=$?
tail -30 gpurun_out/r02_pull_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scratch/msda_bwd_ab.py > gpurun_out/r02_msda_bwd_ab.log 2>&1; rc=$?
cat gpurun_out/r02_msda_bwd_ab.log
exit $rc

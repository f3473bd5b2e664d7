#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "ffn" > gpurun_out/r02_ffn_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02_ffn_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scratch/ffn_check.py 2>&1 | grep "^us" | tee gpurun_out/ffn_check_full.log

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "msda" > gpurun_out/r03_mm_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03_mm_tests.log
[ $rc -eq 0 ] || exit 1
bash scratch/r03_msda_t.sh "DSKD_MSDA_MM=123" | grep "==\|mm_kernel\|bwd_win\|zero_rows\|pull\|total"

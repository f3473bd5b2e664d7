#!/usr/bin/env bash
# matrix-core backward of the coarse levels: parity tests, then per-kernel times with and without it
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "msda" > gpurun_out/r03_mm_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r03_mm_tests.log
[ $rc -eq 0 ] || exit 1
bash scratch/r03_msda_t.sh "DSKD_MSDA_MM=123" "DSKD_MSDA_MM=23" "DSKD_MSDA_MM=0"

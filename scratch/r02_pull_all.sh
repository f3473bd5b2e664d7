#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "msda_bwd" > gpurun_out/r02_pull_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02_pull_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scratch/msda_bwd_ab.py > gpurun_out/r02_msda_bwd_ab_grid.log 2>&1; rc=$?
cat gpurun_out/r02_msda_bwd_ab_grid.log
[ $rc -eq 0 ] || exit $rc
bash dskd_amd/csrc/build.sh -DDSKD_PULL_PROFILE > gpurun_out/r02_pprof_build.log 2>&1 || { tail gpurun_out/r02_pprof_build.log; exit 1; }
AB_PPROF=1 DSKD_MSDA_PULL_LEVELS=01 timeout -k 10 300 python scratch/msda_bwd_ab.py 2>&1 | tee gpurun_out/r02_pull_pprof.log

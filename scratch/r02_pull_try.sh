#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "msda_bwd" > gpurun_out/r02_pull_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02_pull_tests.log
[ $rc -eq 0 ] || exit $rc
for nt in 1024 512 256; do
  for m in 6 5; do
  echo "== threads $nt margin $m"; DSKD_MSDA_PULL_THREADS=$nt AB_MARGIN=$m timeout -k 10 300 python scratch/msda_bwd_ab.py 2>&1 | grep -E "windowed|pull 0 |pull 1 |default"
  done
done

#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for mode in hip hip aten aten; do
  if [ $mode = aten ]; then export DSKD_GN_ATEN=1; else unset DSKD_GN_ATEN; fi
  timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "ffn or group_norm or nchw or graphed_student or full_step" > gpurun_out/flaky_$mode.log 2>&1
  echo "$mode: $(grep -E 'passed|failed' gpurun_out/flaky_$mode.log | tail -1) $(grep -E 'AssertionError: \(' gpurun_out/flaky_$mode.log | head -2 | cut -c1-150)"
done

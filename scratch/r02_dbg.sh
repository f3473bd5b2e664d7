#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -k "out_of_range_keepid" > gpurun_out/r02_dbg_head.log 2>&1; rc=$?
grep -E "^E |passed|failed" gpurun_out/r02_dbg_head.log | cut -c1-300 | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; rc=$?
tail -3 gpurun_out/r02_smoke.log | cut -c1-400
exit $rc

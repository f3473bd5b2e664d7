#!/usr/bin/env bash
# Round-2 GPU check: full -m gpu suite, smoke(), default bench with per-step host / GPU times.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -rxXs > gpurun_out/r02_gpu_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r02_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; rc=$?
tail -3 gpurun_out/r02_smoke.log
[ $rc -eq 0 ] || exit $rc
DSKD_BENCH_STEPTIMES=1 timeout -k 10 400 python bench.py --steps 20 > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; rc=$?
tail -5 gpurun_out/r02_bench_a.err; cat gpurun_out/r02_bench_a.json
exit $rc

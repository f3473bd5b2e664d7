#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
DSKD_GRAPH_TRACE=1 DSKD_BENCH_STEPTIMES=1 timeout -k 10 500 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; rc=$?
grep -E "graph\]|Warning|warn|per-step|Error|fault" gpurun_out/r02_bench_a.err | cut -c1-400 | tail -12; cat gpurun_out/r02_bench_a.json | cut -c1-1500
exit $rc

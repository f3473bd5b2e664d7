#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 10 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r02_bench_prof.json 2> gpurun_out/r02_bench_prof.err; rc=$?
tail -2 gpurun_out/r02_bench_prof.err | cut -c1-300
f=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
python scratch/step_breakdown.py "$f" 70 > gpurun_out/r02_step_breakdown.txt 2>&1
cat gpurun_out/r02_step_breakdown.txt
exit $rc

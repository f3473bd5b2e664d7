#!/usr/bin/env bash
# Kernel breakdown of the last step of the default bench under rocprofv3 --kernel-trace.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 10 --backbone gfl_r50 --no-cpu-baseline --no-mfma-probe > gpurun_out/bdgfl_bench.json 2> gpurun_out/bdgfl_bench.err || { tail -5 gpurun_out/bdgfl_bench.err; exit 1; }
python scratch/step_breakdown.py "$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)" 60 > gpurun_out/bdgfl_step_breakdown.txt 2>&1
tail -48 gpurun_out/bdgfl_step_breakdown.txt

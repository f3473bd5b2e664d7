#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python bench.py --graph --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_graphmode.json 2> gpurun_out/r03_graphmode.err; echo rc=$?
tail -5 gpurun_out/r03_graphmode.err; cat gpurun_out/r03_graphmode.json | cut -c1-400

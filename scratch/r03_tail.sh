#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python scratch/r03_aten_tail.py > gpurun_out/r03_aten_tail.log 2>&1; tail -5 gpurun_out/r03_aten_tail.log

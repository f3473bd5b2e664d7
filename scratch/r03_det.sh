#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
DET=1 timeout -k 10 300 python scratch/r03_determinism.py > gpurun_out/r03_determinism_flag.log 2>&1; tail -40 gpurun_out/r03_determinism_flag.log

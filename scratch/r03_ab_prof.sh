#!/usr/bin/env bash
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
bash scratch/r03_ab.sh DSKD_CONV1X1_LIB
timeout -k 10 400 python scratch/torch_prof.py > gpurun_out/r03_torch_prof.log 2>&1 || tail -5 gpurun_out/r03_torch_prof.log

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fgkd" 2>&1 | tail -5
python scratch/r03_fgkd.py 2>&1 | tail -1

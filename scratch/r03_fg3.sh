#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "fgkd or gfl_distillation or full_step" 2>&1 | tail -3
python scratch/r03_fgkd.py 2>&1 | tail -1

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fused_dense or head_loss_on_gpu or variants or logit_and" > gpurun_out/r03_dl_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r03_dl_tests.log

#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
echo "== T=1000 (ragged)"; T=1000 timeout -k 10 200 python scratch/ffn_check.py 2>&1 | tee gpurun_out/ffn_check_small.log || exit 1
echo "== T=88892"; timeout -k 10 300 python scratch/ffn_check.py 2>&1 | tee gpurun_out/ffn_check_full.log || exit 1

#!/usr/bin/env bash
# A/B: 1x1 conv weight gradients as split-K GEMMs (default) vs MIOpen (DSKD_CONV_WGRAD_MIOPEN=1).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -x -q -k "conv1x1 or full_step" > gpurun_out/r02_convw_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02_convw_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in gemm miopen gemm miopen; do
  if [ $mode = miopen ]; then export DSKD_CONV_WGRAD_MIOPEN=1; else unset DSKD_CONV_WGRAD_MIOPEN; fi
  DSKD_BENCH_STEPTIMES=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-mfma-probe > gpurun_out/r02_convw_$mode.json 2> gpurun_out/r02_convw_$mode.err || exit 1
  echo "$mode: $(python -c "import json;d=json.load(open('gpurun_out/r02_convw_$mode.json'));print(d['value'], d['ms_per_step'])")"
done

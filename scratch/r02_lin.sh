#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "lin256 or tall_linear" > gpurun_out/r02_lin_tests.log 2>&1; rc=$?
tail -12 gpurun_out/r02_lin_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r02_lin_model_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r02_lin_model_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in hip aten hip aten; do
  if [ $mode = aten ]; then export DSKD_LIN256_OFF=1; else unset DSKD_LIN256_OFF; fi
  DSKD_BENCH_STEPTIMES=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-mfma-probe > gpurun_out/r02_lin_bench_$mode.json 2> gpurun_out/r02_lin_bench_$mode.err || { tail -5 gpurun_out/r02_lin_bench_$mode.err; exit 1; }
  echo "$mode: $(python -c "import json;d=json.load(open('gpurun_out/r02_lin_bench_$mode.json'));print(d['value'], d['ms_per_step'])")"
done

#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "ffn" > gpurun_out/r02_ffn_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02_ffn_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r02_ffn_model_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r02_ffn_model_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in block nodes block nodes; do
  if [ $mode = nodes ]; then export DSKD_FFN=nodes; else unset DSKD_FFN; fi
  DSKD_BENCH_STEPTIMES=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-mfma-probe > gpurun_out/r02_ffn_bench_$mode.json 2> gpurun_out/r02_ffn_bench_$mode.err || { tail -5 gpurun_out/r02_ffn_bench_$mode.err; exit 1; }
  echo "$mode: $(python -c "import json;d=json.load(open('gpurun_out/r02_ffn_bench_$mode.json'));print(d['value'], d['ms_per_step'])")"
done

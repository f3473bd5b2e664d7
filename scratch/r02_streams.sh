#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -k "msda" > gpurun_out/r02_streams_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02_streams_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -k "graphed or full_step" > gpurun_out/r02_streams_model.log 2>&1; rc=$?
tail -3 gpurun_out/r02_streams_model.log
[ $rc -eq 0 ] || exit $rc
for mode in multi single multi single; do
  if [ $mode = single ]; then export DSKD_MSDA_BWD_STREAMS=0; else unset DSKD_MSDA_BWD_STREAMS; fi
  DSKD_BENCH_STEPTIMES=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-mfma-probe > gpurun_out/r02_streams_$mode.json 2> gpurun_out/r02_streams_$mode.err || { tail -5 gpurun_out/r02_streams_$mode.err; exit 1; }
  echo "$mode: $(python -c "import json;d=json.load(open('gpurun_out/r02_streams_$mode.json'));print(d['value'], d['ms_per_step'], 'msda_bwd_enc us', d['roofline']['kernels']['msda_bwd_enc']['avg_us'])")"
done

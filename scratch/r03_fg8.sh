#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "fgkd or gfl_distillation or keepid" 2>&1 | tail -4 || exit 1
for v in wave reg; do
  rm -rf gpurun_out/fgprof_$v; mkdir -p gpurun_out/fgprof_$v
  DSKD_FGKD_KL=$v python scratch/r03_fgkd.py 2>&1 | tail -1
  DSKD_FGKD_KL=$v rocprofv3 --kernel-trace --stats -d gpurun_out/fgprof_$v -o fg --output-format csv -- python3 scratch/r03_fgkd.py > /dev/null 2>&1
  python3 - "$(find gpurun_out/fgprof_$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done

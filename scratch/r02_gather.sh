#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "msda" > gpurun_out/r02_gather_tests.log 2>&1; rc=$?
grep -E "^E |passed|failed" gpurun_out/r02_gather_tests.log | cut -c1-250 | tail -8
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp
for mode in default plain; do
  rm -rf /tmp/prof_g
  if [ $mode = plain ]; then export DSKD_MSDA_BWD_GATHER=plain; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python scratch/msda_only.py > /tmp/prof_g.log 2>&1
  f=$(find /tmp/prof_g -name "*kernel_stats.csv" | head -1)
  echo "== gather: $mode"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "msda" in r["Name"] or "zero_rows" in r["Name"]:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
done

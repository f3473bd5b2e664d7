#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "msda_bwd" > gpurun_out/r02_pull_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02_pull_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scratch/msda_bwd_ab.py > gpurun_out/r02_msda_bwd_ab_grid.log 2>&1; rc=$?
cat gpurun_out/r02_msda_bwd_ab_grid.log
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp AB_ONLY=1
rm -rf gpurun_out/prof_pull
DSKD_MSDA_PULL_LEVELS=${PROF_LEVELS:-0123} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pull -o pull -- python scratch/msda_bwd_ab.py > gpurun_out/r02_prof_pull.log 2>&1
f=$(find gpurun_out/prof_pull -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY

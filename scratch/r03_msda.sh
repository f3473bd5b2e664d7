#!/usr/bin/env bash
# round 3: MSDA backward, fused levels-2+3 dot products + statistics by-product; parity tests, then per-kernel times
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "msda" > gpurun_out/r03_msda_tests.log 2>&1 || { tail -40 gpurun_out/r03_msda_tests.log; exit 1; }
tail -2 gpurun_out/r03_msda_tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in new r2; do
  if [ $mode = r2 ]; then export DSKD_MSDA_BWD=r2; else unset DSKD_MSDA_BWD || true; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_msda_$mode -o out --output-format csv -- python3 $R/scratch/msda_only.py > $R/gpurun_out/r03_msda_$mode.log 2>&1 || { tail -5 $R/gpurun_out/r03_msda_$mode.log; exit 1; }
  echo "== $mode"
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r03_msda_$mode/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "msda" in n or "zero_rows" in n or "zero_fill" in n:
        avg = float(r["AverageNs"]) / 1e3
        print(f"{n[:70]:70s} calls={r['Calls']:>4s} avg_us={avg:8.1f}")
        if "fwd" not in n: tot += avg * int(r["Calls"]) / 5
print(f"backward total per call: {tot:.1f} us")
PY
done

#!/usr/bin/env bash
# per-kernel times of the encoder backward under rocprofv3 for a list of environment settings
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1))
  env $v rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_msda_t$i -o out --output-format csv -- python3 $R/scratch/msda_only.py > $R/gpurun_out/r03_msda_t$i.log 2>&1 || { tail -5 $R/gpurun_out/r03_msda_t$i.log; exit 1; }
  echo "== $v"
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r03_msda_t$i/**/*kernel_stats.csv", recursive=True)[0]
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

#!/usr/bin/env bash
# kernel stats of a bench run of another config (whole run, 10 timed + 3 warm-up + probe steps): top kernels by total time
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for bb in gfl_r50 swin_t; do
rm -rf /tmp/prof_$bb
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$bb -o bench -- python3 bench.py --steps 10 --backbone $bb --no-cpu-baseline --no-mfma-probe > gpurun_out/bd_$bb.json 2> gpurun_out/bd_$bb.err || { tail -5 gpurun_out/bd_$bb.err; exit 1; }
python3 - "$(find /tmp/prof_$bb -name '*kernel_stats.csv' | head -1)" $bb <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "naive_conv" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]}: total kernel time {tot/1e6:.1f} ms over the run")
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f} %  calls {r['Calls']:>6s}  avg_us {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:110]}")
PY
done

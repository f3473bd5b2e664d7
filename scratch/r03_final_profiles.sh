#!/usr/bin/env bash
# Round-2 artefacts of the final build: default bench JSON, the same command under rocprofv3 --kernel-trace --stats
# (kernel stats CSV + last-step breakdown), fp32 line.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
DSKD_BENCH_STEPTIMES=1 timeout -k 10 500 python bench.py --steps 20 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || exit 1
grep "per-step" gpurun_out/r03_bench_default.err | cut -c1-160; cut -c1-200 gpurun_out/r03_bench_default.json
rm -rf /tmp/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 10 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_bench_default_under_rocprofv3.json 2> gpurun_out/r03_bench_prof.err || exit 1
python scratch/step_breakdown.py "$(find /tmp/prof_bench -name '*kernel_trace.csv' | head -1)" 70 > gpurun_out/r03_step_breakdown.txt 2>&1
python - "$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "naive_conv" not in r["Name"]]
with open("gpurun_out/r03_bench_default_rocprofv3_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader()
    for r in rows[:160]:
        r["Name"] = r["Name"][:160]; w.writerow(r)
for r in rows:
    if "msda" in r["Name"]:
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
timeout -k 10 500 python bench.py --steps 10 --dtype fp32 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_bench_fp32.json 2> gpurun_out/r03_bench_fp32.err || exit 1
cut -c1-220 gpurun_out/r03_bench_fp32.json
timeout -k 10 400 python bench.py --steps 10 --backbone gfl_r50 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_bench_gfl_r50.json 2> gpurun_out/r03_bench_gfl.err || exit 1
cut -c1-200 gpurun_out/r03_bench_gfl_r50.json
timeout -k 10 500 python bench.py --steps 10 --backbone swin_t --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_bench_swin_t.json 2> gpurun_out/r03_bench_swin.err || exit 1
cut -c1-200 gpurun_out/r03_bench_swin_t.json

#!/usr/bin/env bash
# Artefacts of one round on the final build (run on the GPU box through gpurun): the default bench JSON, the same command
# under rocprofv3 --kernel-trace --stats (kernel stats CSV + last-step breakdown), the fp32 line and the other workloads.
# Usage: tools/prof/round_profiles.sh r04 [quick]     -> gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -uo pipefail
tag=${1:-rXX}; quick=${2:-}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
DSKD_BENCH_STEPTIMES=1 timeout -k 10 500 python bench.py --steps 20 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 1
grep "per-step" gpurun_out/${tag}_bench_default.err | cut -c1-160; cut -c1-200 gpurun_out/${tag}_bench_default.json
rm -rf /tmp/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 10 --no-cpu-baseline --no-mfma-probe > gpurun_out/${tag}_bench_default_under_rocprofv3.json 2> gpurun_out/${tag}_bench_prof.err || exit 1
python tools/prof/step_breakdown.py "$(find /tmp/prof_bench -name '*kernel_trace.csv' | head -1)" 70 > gpurun_out/${tag}_step_breakdown.txt 2>&1
python - "$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)" "$tag" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "naive_conv" not in r["Name"]]
with open(f"gpurun_out/{sys.argv[2]}_bench_default_rocprofv3_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader()
    for r in rows[:160]:
        r["Name"] = r["Name"][:160]; w.writerow(r)
for r in rows:
    if "msda" in r["Name"]:
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
[ -n "$quick" ] && exit 0
timeout -k 10 500 python bench.py --steps 10 --dtype fp32 --no-cpu-baseline --no-mfma-probe > gpurun_out/${tag}_bench_fp32.json 2> gpurun_out/${tag}_bench_fp32.err || exit 1
cut -c1-220 gpurun_out/${tag}_bench_fp32.json
timeout -k 10 400 python bench.py --steps 10 --backbone gfl_r50 --no-cpu-baseline --no-mfma-probe > gpurun_out/${tag}_bench_gfl_r50.json 2> gpurun_out/${tag}_bench_gfl.err || exit 1
cut -c1-200 gpurun_out/${tag}_bench_gfl_r50.json
timeout -k 10 500 python bench.py --steps 10 --backbone swin_t --no-cpu-baseline --no-mfma-probe > gpurun_out/${tag}_bench_swin_t.json 2> gpurun_out/${tag}_bench_swin.err || exit 1
cut -c1-200 gpurun_out/${tag}_bench_swin_t.json

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "fgkd or gfl_distillation or keepid or full_step" 2>&1 | tail -5
python scratch/r03_fgkd.py 2>&1 | tail -1
mkdir -p gpurun_out/fgprof
rocprofv3 --kernel-trace --stats -d gpurun_out/fgprof -o fg --output-format csv -- python3 scratch/r03_fgkd.py 2>&1 | tail -1
f=$(find gpurun_out/fgprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY

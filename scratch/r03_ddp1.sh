#!/usr/bin/env bash
# the production combination on one GPU: 1-rank RCCL process group + DDP + head graphs + teacher graph
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
DSKD_BENCH_DDP1=1 DSKD_BENCH_STEPTIMES=1 timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_ddp1.json 2> gpurun_out/r03_ddp1.err; echo rc=$?
grep -i "warn\|error\|per-step" gpurun_out/r03_ddp1.err | cut -c1-300 | head -12
python -c "
import json; d=json.load(open('gpurun_out/r03_ddp1.json')); print(d['ms_per_step'], d['value'], d['rccl_ranks'], d['config']['execution'], d['config']['final_loss'])"

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graphed or full_step or golden or adamw" 2>&1 | tail -3 || exit 1
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_le.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('colsum level embeds', d['value'], d['ms_per_step'], d['config']['final_loss'])"
DSKD_LEVEL_EMBED_GEMM=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_le.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ones-row GEMM', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done

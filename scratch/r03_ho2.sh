#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "handoff or ffn" 2>&1 | tail -5 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_step or graphed" 2>&1 | tail -5 || exit 1
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_ho_on.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('handoff', d['value'], d['ms_per_step'])"
DSKD_NO_HANDOFF=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_ho_off.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no handoff', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_ho_on.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('handoff', d['value'], d['ms_per_step'])"

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bottleneck" 2>&1 | tail -40 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_step" 2>&1 | tail -5 || exit 1
python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_blk_fused.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['ms_per_step'], d['roofline']['kernels']['msda_bwd_enc']['avg_us'])"
DSKD_BLOCK_UNFUSED=1 python bench.py --steps 20 --warmup 5 --no-mfma-probe --no-cpu-baseline 2>gpurun_out/r03_blk_unfused.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused', d['value'], d['ms_per_step'])"

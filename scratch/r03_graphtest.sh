#!/usr/bin/env bash
# head-only graph-vs-eager test, the coarse whole-step one, the determinism probe, then the op-level torch profile
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "graphed_student_head" > gpurun_out/r03_graphtest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_graphtest.log
tail -5 gpurun_out/r03_graphtest.log
timeout -k 10 300 python scratch/r03_determinism.py > gpurun_out/r03_determinism.log 2>&1; tail -30 gpurun_out/r03_determinism.log
timeout -k 10 400 python scratch/torch_prof.py > gpurun_out/r03_torch_prof.log 2>&1; tail -2 gpurun_out/r03_torch_prof.log

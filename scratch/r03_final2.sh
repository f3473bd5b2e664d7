#!/usr/bin/env bash
# final artefacts: whole -m gpu suite + smoke, then the profile set
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gpu_tests.log
tail -3 gpurun_out/r03_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1; tail -1 gpurun_out/r03_smoke.log
bash scratch/r03_final_profiles.sh

#!/usr/bin/env bash
# final artefacts of round 3: whole -m gpu suite, smoke(), MSDA kernel table + matrix-core phase clocks, then the profile set
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gpu_tests.log
tail -3 gpurun_out/r03_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1; tail -2 gpurun_out/r03_smoke.log
{ echo "# one encoder backward + forward at B=4 bf16, grid-initialised offsets (rocprofv3 --kernel-trace --stats over scratch/msda_only.py)";
  bash scratch/r03_msda_t.sh "DSKD_MSDA_MM=123" "DSKD_MSDA_MM=0";
  echo; echo "# per-phase shader clocks of msda_bwd_mm_kernel (library built with -DDSKD_MM_PROFILE; scratch/r03_mm_prof.py)";
  bash scratch/r03_mm_prof.sh; } > gpurun_out/r03_msda_mm_kernels.txt 2>&1
tail -30 gpurun_out/r03_msda_mm_kernels.txt
bash scratch/r03_final_profiles.sh

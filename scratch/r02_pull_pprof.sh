#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
bash dskd_amd/csrc/build.sh -DDSKD_PULL_PROFILE > gpurun_out/r02_pprof_build.log 2>&1 || { tail gpurun_out/r02_pprof_build.log; exit 1; }
AB_PPROF=1 DSKD_MSDA_PULL_LEVELS=${PROF_LEVELS:-01} timeout -k 10 300 python scratch/msda_bwd_ab.py 2>&1 | tee gpurun_out/r02_pull_pprof.log

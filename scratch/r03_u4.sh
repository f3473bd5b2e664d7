#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
bash scratch/r03_msda_t.sh "DSKD_MSDA_MM=123" "DSKD_HIP_LIB=$PWD/scratch/libs/libdskd_unroll4.so" | grep "==\|fwd_win\|bwd_win\|total"

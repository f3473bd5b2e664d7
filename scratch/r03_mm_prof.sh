#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
DSKD_HIP_LIB=$PWD/scratch/libs/libdskd_mmprof.so timeout -k 10 300 python scratch/r03_mm_prof.py 2>&1 | tail -24

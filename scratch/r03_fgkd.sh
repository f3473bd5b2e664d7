#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
python scratch/r03_fgkd.py 2>&1 | tail -1
for v in 8 16; do DSKD_HIP_LIB=$PWD/scratch/libs/libdskd_rg$v.so python scratch/r03_fgkd.py 2>&1 | tail -1; done

#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for v in loadonly nograd noatom; do echo "== $v"; DSKD_HIP_LIB=$PWD/scratch/libs/libdskd_$v.so python scratch/r03_fgkd.py 2>&1 | tail -1; done

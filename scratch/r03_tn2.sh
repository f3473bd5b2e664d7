#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for v in ns3 ns4 tok32ns4; do echo "== $v"; DSKD_HIP_LIB=$PWD/scratch/libs/libdskd_tn_$v.so python scratch/r03_tn_bench.py 2>&1 | grep "us "; done

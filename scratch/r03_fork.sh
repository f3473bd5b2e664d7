#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
python scratch/r03_fork.py 2>&1 | tail -1
DSKD_TMP_FORK=1 python scratch/r03_fork.py 2>&1 | tail -1
python scratch/r03_fork.py 2>&1 | tail -1
DSKD_TMP_FORK=1 python scratch/r03_fork.py 2>&1 | tail -1

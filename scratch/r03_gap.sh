#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"; python scratch/r03_gap.py 2>&1 | tail -4

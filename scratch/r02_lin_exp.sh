#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for v in "" "-DDSKD_LIN_BUFS=3 -DDSKD_LIN_OCC=3" "-DDSKD_LIN_BUFS=3 -DDSKD_LIN_OCC=2"; do
  bash dskd_amd/csrc/build.sh $v > /tmp/build.log 2>&1 || { tail /tmp/build.log; exit 1; }
  echo "variant [$v]:"; python scratch/lin256_check.py 2>&1 | grep "us lin256" | cut -c1-140
done
bash dskd_amd/csrc/build.sh > /tmp/build.log 2>&1

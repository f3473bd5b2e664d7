#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for v in "" "-DDSKD_FFN_EXPERIMENT_NOSTAGE -DDSKD_FFN_EXPERIMENT_NOREAD" "-DDSKD_FFN_EXPERIMENT_NOREAD" "-DDSKD_FFN_EXPERIMENT_NOSTAGE -DDSKD_FFN_EXPERIMENT_NOEPI" "-DDSKD_FFN_EXPERIMENT_NOSTAGE -DDSKD_FFN_EXPERIMENT_NOEPI -DDSKD_FFN_EXPERIMENT_NOREAD"; do
  bash dskd_amd/csrc/build.sh $v > /tmp/build.log 2>&1 || { tail /tmp/build.log; exit 1; }
  echo "variant [$v]: $(TIME=1 N=1 timeout -k 10 120 python scratch/ffn_only.py 2>&1 | grep '^us')"
done | tee gpurun_out/ffn_exp.txt

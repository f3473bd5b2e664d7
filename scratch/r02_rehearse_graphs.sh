#!/usr/bin/env bash
# Two ranks sharing the one GPU over gloo (the only multi-process set-up available here): the data-parallel path of
# bench.py (DDP wrap, coalesced scalar all-reduce, per-rank teacher stream), graphs off (dist.ranks_share_a_device).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
DSKD_FORCE_GRAPHS=1 DSKD_BENCH_REHEARSE=1 DSKD_BENCH_STEPTIMES=1 DSKD_GRAPH_TRACE=1 timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --batch 2 --steps 5 --warmup 4 --no-mfma-probe --probe-steps 1 > gpurun_out/r02_rehearse_graphs.json 2> gpurun_out/r02_rehearse_graphs.err; rc=$?
grep -E "per-step|stride|Grad strides|graph\]|Error|error" gpurun_out/r02_rehearse_graphs.err | cut -c1-300 | head -8; cut -c1-700 gpurun_out/r02_rehearse_graphs.json
exit $rc

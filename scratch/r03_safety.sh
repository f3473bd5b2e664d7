#!/usr/bin/env bash
# safety runs of the final build: one-rank RCCL + DDP + graphs; 2-rank gloo rehearsal on the shared GPU; a 400-step soak
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
DSKD_BENCH_DDP1=1 DSKD_BENCH_STEPTIMES=1 timeout -k 10 400 python bench.py --steps 20 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_ddp1_final.json 2> gpurun_out/r03_ddp1_final.err || { tail -5 gpurun_out/r03_ddp1_final.err; exit 1; }
cut -c1-330 gpurun_out/r03_ddp1_final.json
DSKD_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 3 --batch 2 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_rehearse_final.json 2> gpurun_out/r03_rehearse_final.err || { tail -5 gpurun_out/r03_rehearse_final.err; exit 1; }
cut -c1-330 gpurun_out/r03_rehearse_final.json
DSKD_BENCH_STEPTIMES=1 timeout -k 10 500 python bench.py --steps 400 --no-cpu-baseline --no-mfma-probe > gpurun_out/r03_soak_final.json 2> gpurun_out/r03_soak_final.err || { tail -5 gpurun_out/r03_soak_final.err; exit 1; }
cut -c1-330 gpurun_out/r03_soak_final.json; grep "allocator" gpurun_out/r03_soak_final.err | cut -c1-250

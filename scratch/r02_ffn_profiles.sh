#!/usr/bin/env bash
# Round-2 artefacts of the fused FFN kernels: timing vs the GEMM chain, PMC counters, ablation builds.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 300 python scratch/ffn_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_ffn_fused_vs_chain.txt || exit 1
cat gpurun_out/r02_ffn_fused_vs_chain.txt | tail -8
bash scratch/r02_ffn_pmc.sh > /dev/null 2>&1; cp gpurun_out/ffn_pmc.txt gpurun_out/r02_ffn_pmc.txt; grep -E "MFMA_BUSY|GRBM|WAVE_CYCLES|WAIT_ANY|FETCH|WRITE" gpurun_out/r02_ffn_pmc.txt
{
echo "# ablation builds of csrc/ffn_mfma.hip (timings only; the variants compute wrong results): us per launch, T = 88 892"
for v in "" "-DDSKD_FFN_EXPERIMENT_NOSTAGE" "-DDSKD_FFN_EXPERIMENT_NOREAD" "-DDSKD_FFN_EXPERIMENT_NOSTAGE -DDSKD_FFN_EXPERIMENT_NOREAD -DDSKD_FFN_EXPERIMENT_NOEPI" "-DDSKD_FFN_RING=16"; do
  bash dskd_amd/csrc/build.sh $v > /tmp/build.log 2>&1 || { tail /tmp/build.log; exit 1; }
  echo "variant [$v]: $(TIME=1 N=1 timeout -k 10 120 python scratch/ffn_only.py 2>&1 | grep '^us')"
done
} | tee gpurun_out/r02_ffn_ablation.txt
bash dskd_amd/csrc/build.sh > /tmp/build.log 2>&1

#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -o p -- python scratch/ffn_only.py > /tmp/pmc_$i.log 2>&1 || { tail -5 /tmp/pmc_$i.log; echo "(pass $i failed)"; }
done
python - <<'PY' | tee gpurun_out/ffn_pmc.txt
import csv, glob, collections
acc, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ffn_fused" not in k: continue
        mode = k.split("ffn_fused_kernel")[1][:8]
        acc[(mode, r["Counter_Name"])] += float(r["Counter_Value"]); n[(mode, r["Counter_Name"])] += 1
for (mode, c) in sorted(acc):
    print(f"{mode:10s} {c:32s} {acc[(mode,c)]/n[(mode,c)]:16.0f}  (n={n[(mode,c)]})")
PY

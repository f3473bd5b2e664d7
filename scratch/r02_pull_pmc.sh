#!/usr/bin/env bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp AB_ONLY=1 DSKD_MSDA_PULL_LEVELS=01
rm -rf gpurun_out/pmc_pull
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_pull -o p1 -- python scratch/msda_bwd_ab.py > gpurun_out/r02_pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_pull -o p2 -- python scratch/msda_bwd_ab.py > gpurun_out/r02_pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_pull/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "msda" not in k: continue
        k = ("pull_R1" if "pull_kernel" in k and "Li1E" in k or ("pull_kernel" in k and ", 1>" in k) else k[:60])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k)
        for c, v in acc[k].items():
            print(f"   {c:26s} {v / n[(k, c)]:.4e} per launch ({n[(k, c)]} launches)")
PY

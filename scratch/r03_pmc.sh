set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 0 2; do
export DSKD_GEMM_LOADER=0 DSKD_GEMM_ABLATE=$a
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_gemm_$a -o out --output-format csv -- python3 $R/scratch/r03_gemm_one.py 16800 256 1024 > $R/gpurun_out/pmc_gemm_$a.log 2>&1 || tail -5 $R/gpurun_out/pmc_gemm_$a.log
done
python3 - <<PY
import csv, glob, collections
for a in (0, 2):
    for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_%d/**/*counter_collection.csv" % a, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_nt" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("ablate", a, {k: sum(v) / len(v) for k, v in acc.items()})
PY

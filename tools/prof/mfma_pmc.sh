#!/usr/bin/env bash
# MFMA-pipe utilisation and LDS bank conflicts per kernel family of the benchmark step (B=4, bf16): rocprofv3 --pmc in two
# separate passes (counters only, no trace domains) over a short bench.py run.
#   pass 1: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE    pass 2: SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE
# Usage (GPU box): bash tools/prof/mfma_pmc.sh r04   ->  gpurun_out/<tag>_mfma_lds_pmc.txt
set -uo pipefail
tag=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/pmc_m /tmp/pmc_l
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -o m -- python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-mfma-probe > /tmp/pmc_m.log 2>&1 || { tail /tmp/pmc_m.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_l -o l -- python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-mfma-probe > /tmp/pmc_l.log 2>&1 || { tail /tmp/pmc_l.log; echo "(LDS pass failed; continuing)"; }
python - "$tag" <<'PY'
import csv, glob, collections, sys, re
def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key); n[k] += 1
    return acc, n
def fam(k):
    k = re.sub(r"^void ", "", k)
    for tag in ("gemm_nt_kernel", "gemm_tn_kernel", "gemm_big_kernel", "ffn_fused_kernel", "lin256_kernel", "msda_bwd_mm", "attn_fwd", "attn_bwd",
                "Cijk", "igemm"):
        if tag in k:
            m = re.search(r"(gemm_nt_kernel<[^>]*>|gemm_tn_kernel<[^>]*>|ffn_fused_kernel<\d>)", k)
            return m.group(1) if m else tag
    return None
m, nm = collect("/tmp/pmc_m")
l, nl = collect("/tmp/pmc_l")
rows = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0, 0.0])
for k, c in m.items():
    f = fam(k)
    if f is None: continue
    rows[f][0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); rows[f][1] += c.get("GRBM_GUI_ACTIVE", 0.0); rows[f][2] += nm[k]
for k, c in l.items():
    f = fam(k)
    if f is None: continue
    rows[f][3] += c.get("SQ_LDS_BANK_CONFLICT", 0.0); rows[f][4] += c.get("SQ_LDS_IDX_ACTIVE", 0.0)
out = open(f"gpurun_out/{sys.argv[1]}_mfma_lds_pmc.txt", "w")
def p(s=""):
    print(s); out.write(s + "\n")
p("MFMA-pipe utilisation and LDS bank conflicts per kernel family over a short bench.py run (B=4, bf16; rocprofv3 --pmc, two passes).")
p("util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of SIMD-cycles with the matrix pipe busy while the")
p("kernel was on the chip (the counter's unit is cycles of one SIMD's pipe, MI355X_MICROARCH.md); conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.")
p(f"{'kernel family':44s} {'launches':>8s} {'MFMA busy Mcyc':>15s} {'chip Mcyc':>10s} {'util':>7s} {'LDS conflict share':>19s}")
for f, (mb, ga, n, bc, ia) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    chip = ga / 8.0
    util = mb / (chip * 1024.0) if chip else float("nan")
    p(f"{f[:44]:44s} {n:8d} {mb / 1e6:15.1f} {chip / 1e6:10.2f} {util:7.3f} {(bc / ia if ia else float('nan')):19.3f}")
PY

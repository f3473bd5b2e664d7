#!/usr/bin/env bash
# HBM traffic of the encoder-shape MSDA launches (B=4, bf16): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
# passes over tools/prof/msda_only.py (MI355X_MICROARCH.md, HBM section), summarised into gpurun_out/<tag>_msda_pmc_hbm_B4_bf16.json
# Usage: tools/prof/msda_pmc.sh r04
set -uo pipefail
export PMC_TAG=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python tools/prof/msda_only.py > /tmp/pmc_f.log 2>&1 || { tail /tmp/pmc_f.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python tools/prof/msda_only.py > /tmp/pmc_w.log 2>&1 || { tail /tmp/pmc_w.log; exit 1; }
rm -rf /tmp/pmc_r
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d /tmp/pmc_r -o r -- python tools/prof/msda_only.py > /tmp/pmc_r.log 2>&1 || { tail /tmp/pmc_r.log; echo "(request-size pass failed; continuing)"; }
python - <<'PY'
import csv, glob, collections, json, os
def collect(d, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            if "msda" not in k and "zero_rows" not in k:
                continue
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}, n
f, nf = collect("/tmp/pmc_f", "FETCH_SIZE")
w, nw = collect("/tmp/pmc_w", "WRITE_SIZE")
def short(k):
    for tag in ("msda_fwd_win", "msda_fwd_kernel", "msda_bwd_pull_apply", "msda_bwd_pull", "msda_bwd_value", "msda_bwd_mm", "msda_bwd_win", "msda_bwd_kernel", "zero_rows"):
        if tag in k:
            extra = ""
            if tag == "msda_bwd_value":
                extra = "_L1" if "Li4ELi12" in k or ", 4, 12" in k else "_L23"
            return tag + extra
    return k[:40]
rq = {c: collect("/tmp/pmc_r", c)[0] for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum")}
out = {"_about": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/prof/msda_only.py: encoder shape "
       "100x167/50x84/25x42/13x21, B=4, bf16, grid-initialised offsets (the benchmark's random-init model); per-launch averages in KB as reported; FETCH_SIZE doubled in "
       "traffic_corrected_MB (gfx950: wide reads are tallied at half, MI355X_MICROARCH.md HBM section), WRITE_SIZE as is.",
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    out["kernels"][short(k)] = {"FETCH_SIZE_KB": round(f.get(k, 0.0), 1), "WRITE_SIZE_KB": round(w.get(k, 0.0), 1), "launches_counted": nf.get(k, 0),
                                "read_requests": {c: round(rq[c].get(k, 0.0)) for c in rq if rq[c]}}
ks = out["kernels"]
# per call of the op: a kernel launched k times per call (msda_bwd_mm: 2, zero_rows: 3) counts k times (5 calls in the run)
fwd = sum((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * v["launches_counted"] / 5 for n, v in ks.items() if n.startswith("msda_fwd")) / 1e3
bwd = sum((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * v["launches_counted"] / 5 for n, v in ks.items() if n.startswith("msda_bwd") or n == "zero_rows") / 1e3
out["traffic_corrected_MB"] = {"msda_fwd_enc": round(fwd, 1), "msda_bwd_enc": round(bwd, 1)}
out["algorithmic_MB"] = {"msda_fwd_enc": 227.56, "msda_bwd_enc": 455.13}
json.dump(out, open(f"gpurun_out/{os.environ['PMC_TAG']}_msda_pmc_hbm_B4_bf16.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

#!/usr/bin/env bash
# VERDICT r3 item 2a, upper bound for the pull kernel: the product build against -DDSKD_PULL_COMPACT (the tile kernel reads a
# compact level- and head-major copy of loc / attn made by pull_compact_kernel).  Kernel times from rocprofv3 --kernel-trace
# --stats over tools/prof/msda_only.py, HBM bytes from --pmc FETCH_SIZE / WRITE_SIZE (separate passes), parity from the
# MSDA backward tests run against the variant.  Usage (on the GPU box): bash tools/prof/msda_pull_compact_ab.sh
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
V=tools/prof/libs/libdskd_pullcompact.so
echo "== parity of the variant (MSDA backward tests against the oracle)"
DSKD_HIP_LIB=$V timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "msda_bwd" 2>&1 | tail -n 2
for name in product variant; do
  lib=""; [ $name = variant ] && lib=$V
  rm -rf /tmp/pc_$name /tmp/pcf_$name /tmp/pcw_$name
  DSKD_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$name -o t -- python tools/prof/msda_only.py > /tmp/pc_$name.log 2>&1 || { tail /tmp/pc_$name.log; exit 1; }
  DSKD_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pcf_$name -o f -- python tools/prof/msda_only.py > /tmp/pcf_$name.log 2>&1 || { tail /tmp/pcf_$name.log; exit 1; }
  DSKD_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pcw_$name -o w -- python tools/prof/msda_only.py > /tmp/pcw_$name.log 2>&1 || { tail /tmp/pcw_$name.log; exit 1; }
done
python - <<'PY'
import csv, glob, collections
def stats(d):
    out = {}
    for f in glob.glob(d + "/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "msda" in n or "pull" in n or "zero_rows" in n:
                out[n[:70]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out
def pmc(d, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("pull" in r["Kernel_Name"]):
                acc[r["Kernel_Name"][:70]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:70]] += 1
    return {k: acc[k] / n[k] for k in acc}
for name in ("product", "variant"):
    print(f"== {name}: kernel, calls, average us | FETCH_SIZE KB (x2 = bytes, gfx950) | WRITE_SIZE KB")
    st, fe, wr = stats(f"/tmp/pc_{name}"), pmc(f"/tmp/pcf_{name}", "FETCH_SIZE"), pmc(f"/tmp/pcw_{name}", "WRITE_SIZE")
    for k, (c, us) in sorted(st.items(), key=lambda kv: -kv[1][1]):
        extra = f" | fetch {fe[k]:10.0f} KB -> {2 * fe[k] / 1e3:7.1f} MB | write {wr.get(k, 0) / 1e3:7.1f} MB" if k in fe else ""
        print(f"  {k:70s} {c:4d} {us:8.1f} us{extra}")
PY

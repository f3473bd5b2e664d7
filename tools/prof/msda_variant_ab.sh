#!/usr/bin/env bash
# Kernel times of the encoder-shape MSDA launches (tools/prof/msda_only.py: B=4, bf16, grid-initialised offsets as in the
# benchmark) for the product library and for variant libraries (tools/prof/libs/libdskd_<name>.so, see build_variant.sh),
# from rocprofv3 --kernel-trace --stats.  Usage (GPU box): bash tools/prof/msda_variant_ab.sh <name> [<name> ...]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
for name in product "$@" product; do
  lib=""; [ "$name" != product ] && lib=tools/prof/libs/libdskd_$name.so
  rm -rf /tmp/va_$name
  DSKD_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/va_$name -o t -- python tools/prof/msda_only.py > /tmp/va_$name.log 2>&1 || { tail /tmp/va_$name.log; exit 1; }
  echo "== $name"
  python - "$(find /tmp/va_$name -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "msda" in n or "zero_rows" in n:
        print(f"  {n[:72]:72s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
done

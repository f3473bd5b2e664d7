#!/usr/bin/env bash
# A/B of one environment switch in the default bench: usage r03_ab.sh VAR [steps]
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
var=$1; steps=${2:-20}
for rep in 1 2; do
  env $var=1 timeout -k 10 500 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_ab_${var}_on_$rep.json 2> gpurun_out/r03_ab_${var}_on_$rep.err
  timeout -k 10 500 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-mfma-probe --probe-steps 0 > gpurun_out/r03_ab_${var}_off_$rep.json 2> gpurun_out/r03_ab_${var}_off_$rep.err
  python - <<PY
import json
for k in ("on", "off"):
    d = json.load(open("gpurun_out/r03_ab_${var}_%s_$rep.json" % k))
    print("$var", k, d["ms_per_step"], d["value"], d["config"]["final_loss"])
PY
done

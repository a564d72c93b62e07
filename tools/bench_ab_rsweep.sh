#!/bin/bash
# A/B of the r-gather row-pattern sweep (sells_rsweep_kernel, GMG_PAT_RSWEEP) on one box: bench.py at $1^3 cells, $2 levels
C=${1:-128}; L=${2:-4}; OUT=${3:-gpurun_out/r03n}
mkdir -p $OUT
for rep in 1 2; do for cfg in "new:GMG_NONE=0" "old:${OLD_ENV:-GMG_PAT_RSWEEP=0}"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 900 python bench.py --cells $C --levels $L --no-cpu-baseline --no-varcoef --no-weak-ref --steps 8 --warmup 2 2>$OUT/rs_${C}_${tag}_$rep.err > $OUT/rs_${C}_${tag}_$rep.json < /dev/null
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/rs_${C}_${tag}_$rep.json").read().strip().splitlines()[-1])
    rc=d["roofline_compressed"]
    print("$C $tag rep$rep | default ms", round(d["ms_per_step"],3), "DoFs/s %.3e" % d["value"], "sweep_us", round(rc["avg_launch_ms"]*1e3,2), "layout frac", round(rc["frac"],3), rc["sweep_signature"], "iters", d["config"]["cg_iterations"], "l2", d["config"]["l2_error_sq"])
except Exception as e:
    print("$C $tag rep$rep FAILED", e)
PY
done; done

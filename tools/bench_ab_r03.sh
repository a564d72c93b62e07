#!/bin/bash
# A/B of the round-3 changes to the SELL-64 / SELL-O sweeps on one box: bench.py at $1^3 cells, $2 levels
#   new = defaults ; old = GMG_NT_ROWWISE=0 GMG_XCD_REMAP_BIG=1 GMG_SELL_DEFER=0 (round-2 behaviour)
C=${1:-256}; L=${2:-5}; OUT=${3:-gpurun_out/r03e}
mkdir -p $OUT
for rep in 1 2; do for cfg in "new:GMG_NONE=0" "old:GMG_NT_ROWWISE=0 GMG_XCD_REMAP_BIG=1 GMG_SELL_DEFER=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 900 python bench.py --cells $C --levels $L --no-cpu-baseline --steps 4 --warmup 1 2>$OUT/ab_${C}_${tag}_$rep.err > $OUT/ab_${C}_${tag}_$rep.json < /dev/null
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/ab_${C}_${tag}_$rep.json").read().strip().splitlines()[-1])
    v=d.get("variable_coefficient",{})
    print("$C $tag rep$rep | generic ms", round(d["ms_per_step_generic"],3), "sweep_us", round(d["roofline"]["avg_launch_ms"]*1e3,1), "frac", round(d["roofline"]["frac"],3),
          "| varcoef ms", round(v.get("ms_per_step",0),3), "sweep_us", round(v["roofline"]["avg_launch_ms"]*1e3,1), "frac", round(v["roofline"]["frac"],3),
          "| default ms", round(d["ms_per_step"],3), "iters", d["config"]["cg_iterations"], d["config"]["cg_iterations_generic"], v.get("cg_iterations"))
except Exception as e:
    print("$C $tag rep$rep FAILED", e)
PY
done; done

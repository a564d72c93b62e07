#!/bin/bash
# default leg of bench.py under a list of environment variants: bash tools/bench_env_sweep.sh <cells> <levels> <outdir> "A=1" "A=2 B=3" ...
C=$1; L=$2; OUT=$3; shift 3
mkdir -p $OUT
i=0
for envs in "GMG_NONE=0" "$@"; do
  i=$((i+1))
  env $envs timeout 900 python bench.py --cells $C --levels $L --no-cpu-baseline --no-varcoef --no-weak-ref --steps 10 --warmup 2 2>$OUT/es_$i.err > $OUT/es_$i.json < /dev/null
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/es_$i.json").read().strip().splitlines()[-1])
    rc=d["roofline_compressed"]
    print("$C | $envs | ms", round(d["ms_per_step"],3), "sweep_us", round(rc["avg_launch_ms"]*1e3,2), rc["sweep_signature"], "iters", d["config"]["cg_iterations"])
except Exception as e:
    print("$C | $envs | FAILED", e)
PY
done

#!/bin/bash
# default leg (+ generic) of bench.py over a list of sizes: bash tools/bench_sizes.sh <outdir> "cells levels" ...
OUT=$1; shift
mkdir -p $OUT
for cl in "$@"; do
  set -- $cl
  timeout 1500 python bench.py --cells $1 --levels $2 --no-cpu-baseline --no-varcoef --no-weak-ref --steps 6 --warmup 2 2>$OUT/sz_$1.err > $OUT/sz_$1.json < /dev/null
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/sz_$1.json").read().strip().splitlines()[-1])
    rc, rg = d["roofline_compressed"], d["roofline"]
    print(f"$1^3 / $2 levels | dofs {d['config']['dofs']:>10d} | default {d['ms_per_step']:8.3f} ms {d['value']:.3e} DoFs/s sweep {rc['avg_launch_ms']*1e3:7.2f} us [{rc['sweep_signature'].split('<')[0]}] | generic {d['ms_per_step_generic']:8.3f} ms {d['value_generic']:.3e} DoFs/s sweep {rg['avg_launch_ms']*1e3:8.2f} us frac {rg['frac']:.3f} | iters {d['config']['cg_iterations']}/{d['config']['cg_iterations_generic']}")
except Exception as e:
    print("$1 FAILED", e)
PY
done

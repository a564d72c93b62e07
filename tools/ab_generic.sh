#!/bin/bash
# A/B of environment variants on the GENERIC (12 B/nnz) leg at any size, one box.
# Usage (GPU box, repo root): bash tools/ab_generic.sh CELLS LEVELS "VAR=1" "VAR=0" ...
CELLS=$1; LEVELS=$2; shift; shift
for e in "$@"; do
  echo "== ${e:-defaults}"
  env $e python3 bench.py --cells $CELLS --levels $LEVELS --legs default,generic,varcoef --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; v=d['variable_coefficient']
print('  generic: ms %.2f sweep us %.1f frac %.3f it %d %s' % (d['ms_per_step_generic'], r['avg_launch_ms']*1e3, r['frac'], d['config']['cg_iterations_generic'], r['sweep_signature']))
print('  varcoef: ms %.2f sweep us %.1f frac %.3f %s' % (v['ms_per_step'], v['roofline']['avg_launch_ms']*1e3, v['roofline']['frac'], v['roofline']['sweep_signature']))"
done

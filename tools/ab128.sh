#!/bin/bash
# A/B of environment variants on the default leg of bench.py at 128^3 only, many steps (resolves ~0.5 %), one box.
# Usage (GPU box, repo root): bash tools/ab128.sh "VAR=1" "VAR=0" ...   (each argument = one variant; "" = defaults)
STEPS=${STEPS:-60}
for e in "$@"; do
  echo -n "== ${e:-defaults}: "
  env $e python3 bench.py --legs default --steps $STEPS --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_compressed']
print('ms %.4f sweep us %.2f it %d' % (d['ms_per_step'], r['avg_launch_ms']*1e3, d['config']['cg_iterations']))"
done

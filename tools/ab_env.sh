#!/bin/bash
# A/B of environment variants on the default legs of bench.py at 128^3 and 288^3, one box.
# Usage (GPU box, repo root): bash tools/ab_env.sh "VAR=1 VAR2=0" "VAR=0" ...   (each argument = one variant; "" = defaults)
for e in "$@"; do
  echo "== ${e:-defaults}"
  env $e python3 bench.py --legs default,weak_ref --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_compressed']; w=d['weak_scaling_ref']
print('  128: ms %.3f sweep us %.2f it %d %s' % (d['ms_per_step'], r['avg_launch_ms']*1e3, d['config']['cg_iterations'], r['sweep_signature']))
print('  288: ms %.2f sweep us %.1f it %d %s' % (w['ms_per_step'], w['roofline_compressed']['avg_launch_ms']*1e3, w['cg_iterations'], w['roofline_compressed']['sweep_signature']))"
done

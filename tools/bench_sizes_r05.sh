#!/bin/bash
# round 5: the default leg at several sizes, z-walk kernels on (default) / off (GMG_PAT_ZWALK=0), one box:  bash tools/bench_sizes_r05.sh "192 5" "256 5" ...
for cl in "$@"; do
  set -- $cl
  for e in "" "GMG_PAT_ZWALK=0"; do
    env $e python3 bench.py --cells $1 --levels $2 --legs default --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_compressed']
print('%4s^3 %-16s ms %.3f  %.3e DoFs/s  it %d  sweep us %.1f  %s' % ('$1', '${e:-default}', d['ms_per_step'], d['value'], d['config']['cg_iterations'], r['avg_launch_ms']*1e3, r['sweep_signature'][:40]))"
  done
done

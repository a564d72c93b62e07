#!/bin/bash
# A/B of the pipelined gather sweep (sells_psweep_kernel) against sells_rsweep_kernel on the default leg, one box.
# Usage (GPU box, repo root): bash tools/ab_pipe.sh [cells levels]
CELLS=${1:-128}; LEVELS=${2:-4}
run() { echo "== $*"; env "$@" python3 bench.py --cells $CELLS --levels $LEVELS --legs default --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_compressed']
print('  ms/solve %.3f  DoFs/s %.3e  sweep us %.2f  iters %d  sig %s' % (d['ms_per_step'], d['value'], r['avg_launch_ms']*1e3, d['config']['cg_iterations'], r['sweep_signature']))"; }
run GMG_PAT_PIPE=0
run GMG_PAT_PIPE=1
run GMG_PAT_PIPE=1 GMG_PAT_FMA=1
run GMG_PAT_PIPE=0
run GMG_PAT_PIPE=1
for w in 1024 1377 1652 2066 2754 4130; do run GMG_PAT_PIPE=1 GMG_PAT_PIPE_WGS=$w; done
run GMG_PAT_PIPE=1 GMG_PAT_STRICT=0
run GMG_PAT_PIPE=1 GMG_PAT_STRICT=0 GMG_PAT_FMA=1

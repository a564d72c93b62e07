#!/bin/bash
# Round 6: A/B of environment variants on the default legs (128^3, 288^3) and the config-3 leg (Q2 256^3) of bench.py, one box.
# Usage (GPU box, repo root): LEGS=default,weak_ref,config3 bash tools/ab_r06.sh "VAR=1" "" ...  (each argument = one variant; "" = defaults)
LEGS=${LEGS:-default,weak_ref,config3}
i=0
for e in "$@"; do
  i=$((i+1))
  echo "== ${e:-defaults}"
  env $e GMG_BENCH_DETAILS=gpurun_out/ab_r06_$i.json python3 bench.py --legs $LEGS --steps ${STEPS:-6} --warmup 2 > /dev/null 2>gpurun_out/ab_r06_$i.err
  python3 - gpurun_out/ab_r06_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print('  128: ms %.3f sweep us %.2f it %d %s' % (d['ms_per_step'], r['avg_launch_ms'] * 1e3, d['config']['cg_iterations'], r.get('sweep_signature')))
w = d.get('weak_scaling_ref')
if isinstance(w, dict) and 'ms_per_step' in w:
    print('  288: ms %.2f sweep us %.1f it %d %s' % (w['ms_per_step'], w['roofline_compressed']['avg_launch_ms'] * 1e3, w['cg_iterations'], w['roofline_compressed']['sweep_signature']))
c = d.get('config3')
if isinstance(c, dict) and 'ms_per_step' in c:
    print('  config3: ms %.1f  A-kernel us %.1f  fgmres it %d  true_rel %.3e  l2 %.3e' % (c['ms_per_step'], c['roofline']['avg_launch_ms'] * 1e3, c['fgmres_iterations'], c['true_residual_rel'], c['l2_error_sq']))
elif c:
    print('  config3:', c)
PY
done

for w in 2048 4096 8192 16384; do
  echo "== PAT_WGS $w"
  GMG_PAT_WGS=$w GMG_SETUP_TIMING=1 python3 bench.py --legs config3 --config3-cells 128 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/c3.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config3']; print('  ms', c['ms_per_step'], 'it', c['fgmres_iterations'], 'kernel us', c['roofline']['avg_launch_ms']*1e3)"
  grep "wide-row tables" /tmp/c3.err | sort | uniq -c | sort -rn | head -4
done

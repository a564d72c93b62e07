for cl in "160 5" "192 5" "224 5"; do set -- $cl; for e in "GMG_PAT_ZWALK=0" "GMG_PAT_ZWALK_T=12" "GMG_PAT_ZWALK_T=6" "GMG_PAT_ZWALK_T=4" "GMG_PAT_ZWALK_T=3"; do env $e python3 bench.py --cells $1 --levels $2 --legs default --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_compressed']
print('%4s^3 %-20s ms %.3f  sweep us %.1f  %s' % ('$1', '$e', d['ms_per_step'], r['avg_launch_ms']*1e3, r['sweep_signature'][-40:]))"; done; done

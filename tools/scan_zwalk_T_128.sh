# 128^3 (headline size): the plane walk with shorter chains (more waves) against the tile sweep -- tools/scan_zwalk_T.sh for the sizes above
for e in "GMG_PAT_ZWALK=0" "GMG_PAT_ZWALK_ROWS=1000000 GMG_PAT_ZWALK_T=12" "GMG_PAT_ZWALK_ROWS=1000000 GMG_PAT_ZWALK_T=6" "GMG_PAT_ZWALK_ROWS=1000000 GMG_PAT_ZWALK_T=4" "GMG_PAT_ZWALK_ROWS=1000000 GMG_PAT_ZWALK_T=3" "GMG_PAT_ZWALK_ROWS=1000000 GMG_PAT_ZWALK_T=2"; do env $e python3 bench.py --cells 128 --levels 4 --legs default --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_compressed']
bv=r.get('by_variant')
print('128^3 %-50s ms %.3f  sweep us %.1f  %s  %s' % ('$e', d['ms_per_step'], r['avg_launch_ms']*1e3, bv, r['sweep_signature'][-40:]))"; done

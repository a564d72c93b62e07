for cfg in "GMG_PAT_WGS=2048" "GMG_PAT_WGS=4096" "GMG_PAT_WGS=8192" "GMG_PAT_WGS=1024" "GMG_SELL_BLOCK=128 GMG_PAT_WGS=4096" "GMG_SELL_BLOCK=512 GMG_PAT_WGS=1024" "GMG_SELL_BLOCK=64 GMG_PAT_WGS=8192"; do
env $cfg timeout 300 python bench.py --cells 128 --no-cpu-baseline --no-varcoef --steps 10 2>/dev/null > gpurun_out/ab.json < /dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").readline())
rc=d.get("roofline_compressed",{})
print("$cfg", "ms/solve", round(d["ms_per_step"],4), "sweep_us", round(rc.get("avg_launch_ms",0)*1e3,2))
PY
done

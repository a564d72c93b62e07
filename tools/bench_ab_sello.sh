for cfg in "GMG_SELL_UN=3" "GMG_SELL_BLOCK=128" "GMG_SELL_BLOCK=512" "GMG_SELL_BLOCK=64" "GMG_NT=0" "GMG_XCD_REMAP=0"; do
env $cfg timeout 300 python bench.py --cells 128 --no-cpu-baseline --steps 4 2>/dev/null > gpurun_out/ab.json < /dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").readline()); v=d["variable_coefficient"]
print("$cfg", "varcoef ms/solve", round(v["ms_per_step"],3), "sweep_us", round(v["roofline"]["avg_launch_ms"]*1e3,2), "| generic sweep_us", round(d["roofline"]["avg_launch_ms"]*1e3,2))
PY
done

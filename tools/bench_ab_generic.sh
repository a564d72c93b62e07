for rep in 1 2; do for cfg in "GMG_NONE=0" "GMG_SELL_BLOCK=128" "GMG_XCD_REMAP=0" "GMG_SELL_BLOCK=128 GMG_XCD_REMAP=0"; do
env $cfg timeout 300 python bench.py --cells 128 --no-cpu-baseline --no-varcoef --steps 4 2>/dev/null > gpurun_out/ab.json < /dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").readline())
print("$cfg", "| generic ms/solve", round(d["ms_per_step_generic"],3), "sweep_us", round(d["roofline"]["avg_launch_ms"]*1e3,2), "| default ms", round(d["ms_per_step"],4))
PY
done; done

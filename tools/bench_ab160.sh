for cfg in "GMG_PERSIST_MAX_SLICES=4096" "GMG_PERSIST=1" "GMG_PERSIST=0"; do
env $cfg timeout 300 python bench.py --cells 160 --levels 4 --no-cpu-baseline --no-varcoef --steps 10 2>/dev/null > gpurun_out/ab.json < /dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").readline())
print("$cfg", "cells 160 ms/solve", round(d["ms_per_step"],4), "iters", d["config"]["cg_iterations"], [ (c["level"], c["rows"], round(c["avg_sweep_ms"]*1e3,2), c["one_launch_per_pass"]) for c in d["roofline"]["coarser_levels"]])
PY
done

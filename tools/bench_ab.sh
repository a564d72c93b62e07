# A/B of environment switches on the default bench leg: bash tools/bench_ab.sh "GMG_PAT_STRICT=1" "GMG_PAT_STRICT=0" ...
for cfg in "$@"; do for cells in 128 256; do
env $cfg timeout 300 python bench.py --cells $cells --no-cpu-baseline --no-varcoef --steps 10 2>/dev/null > gpurun_out/ab.json < /dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").readline())
rc=d.get("roofline_compressed",{})
print("$cfg", "cells", $cells, "ms/solve", round(d["ms_per_step"],4), "sweep_us", round(rc.get("avg_launch_ms",0)*1e3,2), "iters", d["config"]["cg_iterations"], "value %.3e" % d["value"])
PY
done; done

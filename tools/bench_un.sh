for un in 9 27 3; do GMG_SELL_UN=$un python bench.py --no-cpu-baseline --steps 10 2>/dev/null > gpurun_out/bench_un$un.json; python - <<PY
import json
d=json.loads(open("gpurun_out/bench_un$un.json").readline()); v=d["variable_coefficient"]
print("UN", $un, "generic ms", d["ms_per_step_generic"], d["roofline"]["avg_launch_ms"], "varcoef ms", v["ms_per_step"], v["roofline"]["avg_launch_ms"], v["roofline"]["frac"], v["operator_storage"]["layout"], v["cg_iterations"])
PY
done

# A/B of the one-launch smoothing pass (GMG_PERSIST) on the bench configurations; prints ms per solve
for cells in 128 256; do
for p in 0 1; do GMG_PERSIST=$p timeout 300 python bench.py --cells $cells --no-cpu-baseline --no-varcoef --steps 10 2>/dev/null > gpurun_out/bench_persist${p}_$cells.json; python - <<PY
import json
d=json.loads(open("gpurun_out/bench_persist${p}_$cells.json").readline())
print("cells", $cells, "persist", $p, "ms/solve", round(d["ms_per_step"],4), "generic", round(d["ms_per_step_generic"],4), "iters", d["config"]["cg_iterations"], "value", d["value"])
PY
done; done

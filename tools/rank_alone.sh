#!/bin/bash
# kernel time per CG iteration of one rank of an N-rank run, alone on the GPU (tools/rank_alone.py), next to the single-GPU solve of the same
# per-rank problem:  bash tools/rank_alone.sh [cells] [levels] [world] [rank]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ralone
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ralone -o p -- python3 $R/tools/rank_alone.py "$@" > /tmp/ralone.log 2>&1
tail -1 /tmp/ralone.log | cut -c1-400
python3 - <<'PY'
import glob, sqlite3
f = glob.glob("/tmp/ralone/**/*.db", recursive=True)
c = sqlite3.connect(f[0])
rows = c.execute("select start, end, grid_x, name from kernels order by start").fetchall()
# the last 4 solves of 3 iterations: delimit by reduce_post / the norm kernels is fragile here -- take the last 4/5 of the CG kernels by time instead
marks = [i for i, r in enumerate(rows) if "cg_update_kernel" in r[3]]
last = marks[-12:]                                   # 4 solves x 3 iterations
seg = rows[marks[-13] + 1: last[-1] + 1]
tot = sum(e - s for s, e, _, _ in seg) / 1e3
wall = (seg[-1][1] - seg[0][0]) / 1e3
print(f"12 iterations: kernel time {tot / 12:.1f} us per iteration, {wall / 12:.1f} us of stream wall time per iteration (host callbacks inside)")
import collections
agg = collections.OrderedDict()
for s, e, g, n in seg:
    k = (n.split("(")[0][:70], g)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for (n, g), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{cnt / 12:7.1f} x {us / cnt:8.2f} us = {us / 12:8.1f} us per iteration | {g:9d} | {n}")
PY

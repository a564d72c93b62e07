#!/bin/bash
# per-kernel table (rocprofv3 --kernel-trace) of any bench.py invocation:  bash tools/kstats.sh [ENV=..] -- <bench.py args>
ENVS=()
while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
env "${ENVS[@]}" timeout -k 5 900 rocprofv3 --kernel-trace -d /tmp/kst -o p -- python3 $ROOTDIR/bench.py "$@" > /tmp/kst.log 2>&1 < /dev/null
python3 - <<'PY'
import glob, sqlite3
f = glob.glob("/tmp/kst/**/*.db", recursive=True)
c = sqlite3.connect(f[0])
rows = c.execute("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name, grid_x order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"total kernel time {tot/1e3:.3f} ms")
for r in rows[:14]:
    print(f"{r[3]:6d} {r[4]:10.1f} {r[5]:8.2f} {100*r[4]/tot:5.1f}% | {r[1]:8d} {r[2]:4d} | {r[0][:110]}")
PY

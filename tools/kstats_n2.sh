#!/bin/bash
# per-kernel table of rank 0 of a 2-rank bench.py run on ONE GPU (host transport): rank 1 runs plainly, rank 0 under rocprofv3
#   bash tools/kstats_n2.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GMG_SHARE_GPU=1 GMG_TRANSPORT=host OMP_NUM_THREADS=2 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 LOCAL_WORLD_SIZE=2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kn2
RANK=1 LOCAL_RANK=1 timeout -k 5 900 python3 $R/bench.py --gpus 2 --no-cpu-baseline "$@" > /tmp/kn2_r1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 5 900 rocprofv3 --kernel-trace -d /tmp/kn2 -o p -- python3 $R/bench.py --gpus 2 --no-cpu-baseline "$@" > /tmp/kn2_r0.log 2>&1 < /dev/null
wait $P1
python3 - <<'PY'
import glob, sqlite3
f = glob.glob("/tmp/kn2/**/*.db", recursive=True)
c = sqlite3.connect(f[0])
rows = c.execute("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name, grid_x order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"total kernel time {tot/1e3:.3f} ms")
for r in rows[:26]:
    print(f"{r[3]:6d} {r[4]:10.1f} {r[5]:8.2f} {100*r[4]/tot:5.1f}% | {r[1]:8d} {r[2]:4d} | {r[0][:110]}")
PY
tail -2 /tmp/kn2_r0.log | cut -c1-300

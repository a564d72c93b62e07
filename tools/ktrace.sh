#!/bin/bash
# Usage (GPU box): bash tools/ktrace.sh <tag> [ENV=VAL ...] -> per-kernel table of one 128^3 solve loop (tools/tune.py child)
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TUNE_VARIANTS='[{}]'
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/trace -o p -- python3 $ROOTDIR/tools/tune.py child > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY' > $OUT/summary.txt 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)
c = sqlite3.connect(f[0])
rows = c.execute("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name, grid_x order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"total kernel time {tot/1e3:.3f} ms")
for r in rows[:45]:
    print(f"{r[3]:6d} {r[4]:10.1f} {r[5]:8.2f} {100*r[4]/tot:5.1f}% | {r[1]:8d} {r[2]:4d} | {r[0][:90]}")
PY
rm -rf $OUT/trace
cat $OUT/summary.txt

#!/bin/bash
# Usage (GPU box): bash tools/ktrace_q2.sh [cells levels] -> per-kernel table of the Q2 patch-smoother FGMRES run
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/kt_q2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${3:-600} rocprofv3 --kernel-trace -d $OUT/trace -o p -- python3 $ROOTDIR/tools/config3.py --cells ${1:-64} --levels ${2:-4} --steps 1 --warmup 1 > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY' > $OUT/summary.txt 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)
c = sqlite3.connect(f[0])
rows = c.execute("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name, grid_x order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"total kernel time {tot/1e3:.3f} ms")
for r in rows[:40]:
    print(f"{r[3]:6d} {r[4]:10.1f} {r[5]:8.2f} {100*r[4]/tot:5.1f}% | {r[1]:8d} {r[2]:4d} | {r[0][:100]}")
PY
rm -rf $OUT/trace
tail -2 $OUT/trace.log | cut -c1-400
cat $OUT/summary.txt

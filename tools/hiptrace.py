#!/usr/bin/env python3
"""Host API calls and kernels of the tail of a rocprofv3 --kernel-trace --hip-trace database, merged by time:
python tools/hiptrace.py <dir> [nkernels]   -- the window of the last nkernels kernels; shows which host call sits in front of a gap on the GPU."""
import glob, os, sqlite3, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
con = sqlite3.connect(f[0])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
for s, e, g, n in con.execute("select start, end, grid_x, name from kernels order by start"):
    ev.append((s, e, "GPU ", f"{n[:70]} [{g}]"))
reg = [t for t in tabs if t.startswith("regions") or t == "regions"]
q = None
for t in ("regions", "regions_and_samples"):
    if t in tabs:
        q = t
        break
if q:
    cols = [r[1] for r in con.execute(f"pragma table_info({q})")]
    for r in con.execute(f"select start, end, name from {q} order by start"):
        ev.append((r[0], r[1], "HOST", str(r[2])[:70]))
ev.sort()
ks = [e for e in ev if e[2] == "GPU "]
lo, hi = ks[-min(count, len(ks))][0], ks[-1][1]
ev = [e for e in ev if lo <= e[0] <= hi]
t0 = ev[0][0]
for s, e, k, n in ev:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.2f} {k} {n}")

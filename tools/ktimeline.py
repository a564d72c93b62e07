#!/usr/bin/env python3
"""Ordered kernel timeline of a rocprofv3 --kernel-trace database: python tools/ktimeline.py <dir> [first] [count]
prints start offset (us), duration, gap to the previous kernel's end, grid, name -- one CG iteration of the bench at a glance."""
import glob, os, sqlite3, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
con = sqlite3.connect(f[0])
rows = con.execute("select start, end, grid_x, name from kernels order by start").fetchall()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
if first < 0:
    first = max(0, len(rows) + first)
t0 = rows[first][0]
prev_end = rows[first - 1][1] if first > 0 else rows[first][0]
print(f"# {len(rows)} kernels; showing {first}..{first + count}")
for s, e, g, n in rows[first:first + count]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.2f} gap {(s - prev_end) / 1e3:7.2f} | {g:9d} | {n[:80]}")
    prev_end = e

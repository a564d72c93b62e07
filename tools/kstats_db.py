#!/usr/bin/env python3
"""Per kernel and grid size table of a rocprofv3 --kernel-trace database (rocpd .db): python tools/kstats_db.py <dir> [n]"""
import glob, os, sqlite3, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
if not f:
    sys.exit("no .db under " + sys.argv[1])
con = sqlite3.connect(f[0])
rows = con.execute("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 "
                   "from kernels group by name, grid_x order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"# total kernel time {tot/1e3:.3f} ms ; calls total_us avg_us min_us pct | grid wg | name")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[3]:6d} {r[4]:11.1f} {r[5]:9.2f} {r[6]:9.2f} {100*r[4]/tot:5.1f}% | {r[1]:8d} {r[2]:4d} | {r[0][:100]}")

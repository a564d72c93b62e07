#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd databases (gpurun_out/prof_<tag>/{trace,fetch,write}/*.db)
into the text files committed under profiles/.

    python profiles/summarize.py gpurun_out/prof_r01 profiles/r01

Writes <out>_kernel_stats.txt (the `--kernel-trace --stats` view: per kernel and per
grid size, so multigrid levels are told apart) and <out>_hbm_traffic.txt / .json
(FETCH_SIZE / WRITE_SIZE per launch of the dominant kernel, separate --pmc passes).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read -> doubled before use; units are KiB."""
import glob
import json
import os
import sqlite3
import sys


def db(path):
    f = glob.glob(os.path.join(path, "*.db"))
    return sqlite3.connect(f[0]) if f else None


def main():
    src, out = sys.argv[1], sys.argv[2]
    lines = []
    con = db(os.path.join(src, "trace"))
    rows = con.execute(
        "select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3, "
        "min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name, grid_x order by 5 desc").fetchall()
    tot = sum(r[4] for r in rows)
    lines.append(f"# rocprofv3 --kernel-trace --stats : python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline")
    lines.append(f"# total kernel time {tot/1e3:.3f} ms ; columns: calls total_us avg_us min_us max_us pct | grid wg vgpr sgpr lds")
    for r in rows:
        lines.append(f"{r[3]:6d} {r[4]:12.1f} {r[5]:9.2f} {r[6]:9.2f} {r[7]:9.2f} {100*r[4]/tot:5.1f}% | "
                     f"{r[1]:8d} {r[2]:4d} {r[8]:4d} {r[9]:4d} {r[10]:6d} | {r[0][:100]}")
    open(out + "_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))

    # The bench command runs the product path (steps+warmup solves) and then a shorter pass with the
    # operator compression switched off (roofline_generic).  Dominant kernel of the PRODUCT path = the fused
    # sweep (EPI_SWEEP = 3) with the most launches; the generic pass' sweep is reported next to it.
    sw = con.execute("select name, grid_x, avg(end-start)/1e3, count(*), sum(end-start) from kernels "
                     "where name like '%_kernel<3,%' group by name, grid_x order by count(*) desc, sum(end-start) desc").fetchall()
    tl, res = [], {}
    picks = [("", sw[0])]
    for r in sw[1:]:
        if r[0] != sw[0][0] and r[4] == max(q[4] for q in sw if q[0] != sw[0][0]):
            picks.append(("generic_", r))
    for prefix, dom in picks:
        res[prefix + "kernel"] = dom[0]; res[prefix + "grid_x"] = dom[1]
        res[prefix + "avg_us_kernel_trace"] = dom[2]; res[prefix + "launches"] = dom[3]
        tl.append(f"# {'product' if not prefix else 'uncompressed (roofline_generic) pass'}: {dom[0]} grid_x={dom[1]} avg {dom[2]:.2f} us over {dom[3]} launches (kernel-trace pass)")
        for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            c = db(os.path.join(src, sub))
            if c is None:
                continue
            r = c.execute("select avg(value), min(value), max(value), count(*), avg(end-start)/1e3 from counters_collection "
                          "where counter_name=? and kernel_name=? and grid_size_x=?", (cname, dom[0], dom[1])).fetchone()
            if r and r[3]:
                res[prefix + cname + "_KiB_avg"] = r[0]
                res[prefix + cname + "_launches"] = r[3]
                res[prefix + cname + "_pass_avg_us"] = r[4]
                tl.append(f"{cname}: avg {r[0]:.1f} KiB (min {r[1]:.1f} max {r[2]:.1f}) over {r[3]} launches, avg {r[4]:.2f} us in that pass")
        if prefix + "FETCH_SIZE_KiB_avg" in res and prefix + "WRITE_SIZE_KiB_avg" in res:
            fetch = 2.0 * res[prefix + "FETCH_SIZE_KiB_avg"] * 1024.0     # gfx950: x2 for wide coalesced streams
            write = res[prefix + "WRITE_SIZE_KiB_avg"] * 1024.0
            res[(prefix or "sweep_") + "hbm_bytes_per_launch"] = fetch + write
            res[prefix + "fetch_bytes_corrected"] = fetch
            res[prefix + "write_bytes"] = write
            tl.append(f"HBM traffic per launch = 2*FETCH_SIZE + WRITE_SIZE = {fetch/1e6:.1f} MB + {write/1e6:.1f} MB = {(fetch+write)/1e6:.1f} MB")
    open(out + "_hbm_traffic.txt", "w").write("\n".join(tl) + "\n")
    json.dump(res, open(out + "_hbm_traffic.json", "w"), indent=1)
    print("\n".join(tl))


if __name__ == "__main__":
    main()

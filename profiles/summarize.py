#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd databases (gpurun_out/prof_<tag>/{trace,fetch,write}/*.db) into the text files
committed under profiles/.

    python profiles/summarize.py gpurun_out/prof_r02 profiles/r02 [--cells 128 --levels 4 --cmd "..."]

Writes <out>_kernel_stats.txt (the `--kernel-trace --stats` view: per kernel and per grid size, so multigrid levels
are told apart) and <out>_hbm_traffic.txt / .json (FETCH_SIZE / WRITE_SIZE per launch of the finest-level fused sweep
kernels, separate --pmc passes) and refreshes profiles/traffic_latest.json, which bench.py attaches as
`roofline.traffic` only when kernel family / problem / row count match its own run.

Counter corrections (MI355X_MICROARCH.md, HBM section): units are KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced streaming read.  Both factors are CALIBRATED in the same passes on `stream_copy_kernel`
(gmg_stream_probe: exactly 2^30 B read + 2^30 B written per launch) and the calibrated factors are what is applied;
the guide's x2 / x1 are the fallback when the probe kernel is absent."""
import argparse
import glob
import json
import os
import sqlite3


def db(path):
    f = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def family(name):
    if any("gmg::" + k + "<" in name for k in ("sells_sweep_kernel", "sells_rsweep_kernel", "sells_r2sweep_kernel", "sells_tsweep_kernel", "sells_zsweep_kernel")):   # the fused sweeps on the shared-offset pattern table (XM = 0/1/2 variants)
        return "sells_kernel"
    for fam in ("sells_kernel", "sellp_kernel", "sellc_kernel", "sello_kernel", "sell_kernel", "csr_stream1_kernel"):
        if "gmg::" + fam + "<" in name:
            return fam
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--cells", type=int, default=128)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--cmd", default="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline")
    ap.add_argument("--no-latest", action="store_true")
    ap.add_argument("--merge-latest", action="store_true", help="add this run's records to profiles/traffic_latest.json (same kernels.hpp sha) instead of replacing it")
    ap.add_argument("--order", type=int, default=1, help="2: the config-3 leg (Q2): the record is the wide-row operator mat-vec r -= A dx of the patch sweep; "
                                                         "3: the config-5 leg (2-D Stokes velocity block, --cells = cells per direction): the same kernel family on 2 (2 cells - 1)^2 rows")
    ap.add_argument("--bench-log", default=None, help="stdout of the profiled bench.py run (default: <src>/trace.log): its JSON line supplies the sweep signatures")
    a = ap.parse_args()
    src, out = a.src, a.out
    lines = []
    con = db(os.path.join(src, "trace"))
    rows = con.execute(
        "select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3, "
        "min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name, grid_x order by 5 desc").fetchall()
    tot = sum(r[4] for r in rows)
    lines.append(f"# rocprofv3 --kernel-trace --stats : {a.cmd}")
    lines.append(f"# total kernel time {tot/1e3:.3f} ms ; columns: calls total_us avg_us min_us max_us pct | grid wg vgpr sgpr lds")
    for r in rows:
        lines.append(f"{r[3]:6d} {r[4]:12.1f} {r[5]:9.2f} {r[6]:9.2f} {r[7]:9.2f} {100*r[4]/tot:5.1f}% | "
                     f"{r[1]:8d} {r[2]:4d} {r[8]:4d} {r[9]:4d} {r[10]:6d} | {r[0][:110]}")
    open(out + "_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))

    cdb = {c: db(os.path.join(src, sub)) for c, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"))}

    min_frac = 0.0

    def counter(cname, kname, grid):
        c = cdb[cname]
        if c is None:
            return None
        mx = c.execute("select max(end-start) from counters_collection where counter_name=? and kernel_name=? and grid_size_x=?", (cname, kname, grid)).fetchone()[0]
        if mx is None:
            return None
        r = c.execute("select avg(value), min(value), max(value), count(*), avg(end-start)/1e3 from counters_collection "
                      "where counter_name=? and kernel_name=? and grid_size_x=? and (end-start) >= ?", (cname, kname, grid, min_frac * mx)).fetchone()
        return r if r and r[3] else None

    tl = []
    # ---- calibration on the copy probe (exact byte counts) ----
    ffac, wfac, cal = 2.0, 1.0, "guide defaults (x2 FETCH_SIZE, x1 WRITE_SIZE): probe kernel not found"
    pk = con.execute("select name, grid_x, count(*) from kernels where name like '%stream_copy_kernel%' group by name, grid_x").fetchall()
    if pk:
        f = counter("FETCH_SIZE", pk[0][0], pk[0][1])
        w = counter("WRITE_SIZE", pk[0][0], pk[0][1])
        exact = float(1 << 30)
        if f and w:
            ffac, wfac = exact / (f[0] * 1024.0), exact / (w[0] * 1024.0)
            cal = (f"calibrated on stream_copy_kernel (2^30 B read + 2^30 B written per launch): FETCH_SIZE avg {f[0]:.0f} KiB -> x{ffac:.3f}, "
                   f"WRITE_SIZE avg {w[0]:.0f} KiB -> x{wfac:.3f}")
    tl.append("# counter corrections: " + cal)

    # ---- finest-level fused sweeps (EPI_SWEEP = 3 kernels and the sells_sweep_kernel variants): largest grid per family;
    #      variants of one family at that grid (x updated every second sweep) are averaged, weighted by launches ----
    sw = con.execute("select name, grid_x, avg(end-start)/1e3, count(*) from kernels where name like '%_kernel<3,%' or name like '%sells_sweep_kernel<%' or name like '%sells_rsweep_kernel<%' or name like '%sells_r2sweep_kernel<%' or name like '%sells_tsweep_kernel<%' "
                     "or name like '%sells_zsweep_kernel<%, 3>%' "      # (the z-walk sweep: EPI_SWEEP = 3 is its last template argument; its mat-vec forms are not sweeps)
                     "group by name, grid_x order by grid_x desc").fetchall()
    recs = []
    nrows = (a.cells - 1) ** 3
    if a.order in (2, 3):
        # config-3 leg: the timed kernel is the operator mat-vec of the patch sweep, sells_kernel<EPI_SUB = 1, ..., K = 5, VD, ..., WL> on the
        # finest Q2 level ((2 cells - 1)^3 rows); the additive-Schwarz mat-vec (EPI_ADDTO = 4) runs on the same grid and is listed beside it
        nrows = (2 * a.cells - 1) ** 3 if a.order == 2 else 2 * (2 * a.cells - 1) ** 2
        # (coarser levels launch the same kernels on the same capped grid: only the dispatches of the finest level -- at least 0.75 of the
        # longest -- enter the record)
        sw = []
        zw = False
        for epi in ("1", "4"):
            # round 5: levels of >= 3.5e6 rows run the z-walk form (sellw_zwalk_kernel<EPI>); smaller ones sells_kernel<EPI, ..., WL>
            rows = con.execute("select name, grid_x, max(end-start) from kernels where name like ? group by name, grid_x order by 3 desc",
                               ("%gmg::sellw_zwalk_kernel<" + epi + ",%",)).fetchall()
            if not rows:
                rows = con.execute("select name, grid_x, max(end-start) from kernels where name like ? group by name, grid_x order by 3 desc",
                                   ("%gmg::sells_kernel<" + epi + ",%, 5, true%",)).fetchall()
            if rows:
                name, grid, mx = rows[0]
                if "sellw_zwalk_kernel" in name:
                    # the z-walk kernels launch one workgroup per four chains: every level has its own grid size -- all launches of the
                    # largest grid are the finest level (no duration filter: one slow outlier would otherwise hide all the others)
                    zw = True
                    r = con.execute("select avg(end-start)/1e3, count(*) from kernels where name=? and grid_x=?", (name, grid)).fetchone()
                else:
                    r = con.execute("select avg(end-start)/1e3, count(*) from kernels where name=? and grid_x=? and (end-start) >= ?", (name, grid, 0.75 * mx)).fetchone()
                sw.append((name, grid, r[0], r[1]))
        min_frac = 0.0 if zw else 0.75
    fams = {}
    for name, grid, avg_us, cnt in sw:
        fam = family(name)
        if a.order in (2, 3):
            fam = "sells_kernel_wide" if ("sells_kernel<1," in name or "sellw_zwalk_kernel<1," in name) else "sells_kernel_wide_schwarz"
        if fam is None:
            continue
        # one record per (family, grid): the finest level and, when its batched variant runs on a smaller grid than the next
        # level's (NB = 2 on 1024 workgroups), that level too; small levels (< 10^5 threads) are left out
        if grid < 100000:
            continue
        key = (fam, "tile" if "sells_tsweep_kernel<" in name else ("nt" if ", true>" in name and fam == "sell_kernel" else ""), grid)   # (the tile sweep of the finest level and the gather sweep of the next one can share a grid size)
        if key not in fams:
            fams[key] = dict(grid=grid, members=[])
        fams[key]["members"].append((name, avg_us, cnt))
    for (fam, _, _g), info in fams.items():
        grid = info["grid"]
        tot_cnt = sum(m[2] for m in info["members"])
        avg_us = sum(m[1] * m[2] for m in info["members"]) / tot_cnt
        names = [m[0] for m in info["members"]]
        rec = dict(family=fam, kernel=" | ".join(names), grid_x=grid, avg_us_kernel_trace=avg_us, launches=tot_cnt, cells=a.cells, levels=a.levels, rows=nrows, order=a.order)
        tl.append(f"# {' | '.join(names)} grid_x={grid}: avg {avg_us:.2f} us over {tot_cnt} launches (kernel-trace pass)")
        fsum = wsum = n_f = n_w = 0.0
        for name, _, _ in info["members"]:
            f = counter("FETCH_SIZE", name, grid)
            w = counter("WRITE_SIZE", name, grid)
            if f and w:
                fsum += f[0] * f[3]; n_f += f[3]; wsum += w[0] * w[3]; n_w += w[3]
        if n_f and n_w:
            fk, wk = fsum / n_f, wsum / n_w
            fb, wb = ffac * fk * 1024.0, wfac * wk * 1024.0
            rec.update(FETCH_SIZE_KiB_avg=fk, WRITE_SIZE_KiB_avg=wk, fetch_bytes_corrected=fb, write_bytes_corrected=wb,
                       hbm_bytes_per_launch=fb + wb, fetch_factor=ffac, write_factor=wfac)
            tl.append(f"  FETCH_SIZE avg {fk:.1f} KiB, WRITE_SIZE avg {wk:.1f} KiB (launch-weighted over the variants)")
            tl.append(f"  HBM traffic per launch = {ffac:.3f}*FETCH + {wfac:.3f}*WRITE = {fb/1e6:.1f} MB + {wb/1e6:.1f} MB = {(fb+wb)/1e6:.1f} MB "
                      f"-> {(fb+wb)/avg_us/1e3:.0f} GB/s at the kernel-trace duration")
        recs.append(rec)
    # sweep signatures as the library reported them in the profiled run (gmg_sweep_signature, printed by bench.py), and the sha of
    # the kernel source: bench.py attaches these measurements only to runs of the same kernels
    sigs = {}
    try:
        det = os.path.join(src, "legs_trace.json")
        if os.path.exists(det):                                     # round 6: everything the run measured (bench.py's details file)
            bj = json.load(open(det))
            legs = (("generic", bj.get("roofline_generic")), ("default", bj.get("roofline")), ("varcoef", (bj.get("variable_coefficient") or {}).get("roofline")))
        else:
            blog = a.bench_log or os.path.join(src, "trace.log")
            line = [ln for ln in open(blog).read().splitlines() if ln.startswith("{") and '"roofline"' in ln][-1]
            bj = json.loads(line)
            legs = (("generic", bj.get("roofline")), ("default", bj.get("roofline_compressed")), ("varcoef", (bj.get("variable_coefficient") or {}).get("roofline")))
        for fam_leg, blk in legs:
            if blk and blk.get("sweep_signature"):
                sigs[blk["sweep_signature"].split("<")[0].replace("sells_sweep_kernel", "sells_kernel").replace("sells_rsweep_kernel", "sells_kernel").replace("sells_r2sweep_kernel", "sells_kernel").replace("sells_tsweep_kernel", "sells_kernel").replace("sells_zsweep_kernel", "sells_kernel")] = blk["sweep_signature"]
    except Exception as e:
        tl.append(f"# (no sweep signatures: {e})")
    for rec in recs:
        rec["signature"] = sigs.get(rec["family"])
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256(open(os.path.join(root, "gridapsolvers.jl_amd", "csrc", "kernels.hpp"), "rb").read()).hexdigest()[:16]
    recs.sort(key=lambda r: -r["avg_us_kernel_trace"])      # per family the finest level (the slowest launch) comes first: bench.py takes the first match
    res = dict(tag=os.path.basename(out), command=a.cmd, calibration=cal, kernels_hpp_sha=sha, kernels=recs)
    open(out + "_hbm_traffic.txt", "w").write("\n".join(tl) + "\n")
    json.dump(res, open(out + "_hbm_traffic.json", "w"), indent=1)
    latest = os.path.join(os.path.dirname(os.path.abspath(out)), "traffic_latest.json")
    if a.merge_latest:
        try:
            old = json.load(open(latest))
        except Exception:
            old = None
        if old and old.get("kernels_hpp_sha") == sha:
            keep = [r for r in old.get("kernels", []) if not any(r.get("family") == n.get("family") and r.get("cells") == n.get("cells") and r.get("levels") == n.get("levels")
                                                                   and r.get("order", 1) == n.get("order", 1) for n in recs)]
            res = dict(old, tag=old.get("tag", "") + "+" + os.path.basename(out), kernels=keep + recs)
        json.dump(res, open(latest, "w"), indent=1)
    elif not a.no_latest:
        json.dump(res, open(latest, "w"), indent=1)
    print("\n".join(tl))


if __name__ == "__main__":
    main()

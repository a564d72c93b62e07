#!/bin/bash
# HBM-side traffic (TCC_EA: FETCH_SIZE / WRITE_SIZE, separate --pmc passes as the MI355X guide prescribes) per launch of one kernel
# family in any command:   bash tools/pmc_traffic.sh <kernel-name-substring> <min grid_x> [ENV=.. ENV=..] -- <python script + args>
# Units: KiB; FETCH_SIZE is calibrated in the same pass on stream_copy_kernel (exactly 2^30 B read + 2^30 B written per launch) when
# the command runs gmg_stream_probe (bench.py's default leg does); otherwise the guide's x2 (gfx950) / x1 apply.
PAT=$1; MING=$2; shift 2
ENVS=()
while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$C
  env "${ENVS[@]}" timeout -k 5 1200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmct_$C -o b -- python3 "$@" > /tmp/pmct_$C.log 2>&1 < /dev/null
done
python3 - "$PAT" "$MING" <<'PY'
import glob, sqlite3, sys
pat, ming = sys.argv[1], int(sys.argv[2])
res = {}
cal = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmct_{C}/**/*.db", recursive=True)
    if not f:
        print("no db for", C); print(open(f"/tmp/pmct_{C}.log").read()[-1500:]); continue
    con = sqlite3.connect(f[0])
    rows = con.execute("select kernel_name, grid_size_x, avg(value), count(*) from counters_collection where counter_name = ? "
                       "group by kernel_name, grid_size_x", (C,)).fetchall()
    for name, grid, val, n in rows:
        if "stream_copy_kernel" in name:
            cal[C] = (1 << 30) / (val * 1024.0)
        if pat in name and grid >= ming:
            res.setdefault((name, grid), {})[C] = (val, n)
fc, wc = cal.get("FETCH_SIZE", 2.0), cal.get("WRITE_SIZE", 1.0)
print(f"calibration on stream_copy_kernel: FETCH_SIZE x {fc:.3f}, WRITE_SIZE x {wc:.3f}" + ("" if cal else "  (probe kernel absent: guide factors)"))
for (name, grid), d in sorted(res.items(), key=lambda kv: -kv[0][1]):
    f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
    print(f"{name[:96]:96s} grid {grid:9d} launches {f[1]:5d}  read {f[0]*1024*fc/1e6:10.1f} MB  written {w[0]*1024*wc/1e6:10.1f} MB  total {(f[0]*fc+w[0]*wc)*1024/1e6:10.1f} MB per launch")
PY

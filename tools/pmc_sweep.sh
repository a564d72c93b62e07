#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_sweep.sh <tag> [ENV=VAL ...]   -> gpurun_out/pmc_<tag>/summary.txt
# Counter passes on the 128^3 solve (tools/tune.py child, default variant); one counter group per pass.
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TUNE_VARIANTS='[{}]'
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o p -- python3 $ROOTDIR/tools/tune.py child > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
python3 - "$OUT" <<'PY' > $OUT/summary.txt 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
for sub in ("fetch", "write", "sq1", "sq2", "tcc"):
    f = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    if not f:
        print(sub, "no db; log tail:"); print(open(os.path.join(out, sub + ".log")).read()[-600:]); continue
    c = sqlite3.connect(f[0])
    try:
        rows = c.execute("select kernel_name, grid_size_x, counter_name, avg(value), count(*), avg(end-start)/1e3 from counters_collection "
                         "group by kernel_name, grid_size_x, counter_name order by sum(end-start) desc").fetchall()
    except Exception as e:
        print(sub, "query failed", e); continue
    seen = {}
    for k, g, cn, v, n, us in rows:
        key = (k, g)
        if key not in seen:
            if len(seen) >= 4: continue
            seen[key] = True
            print(f"[{sub}] {k[:70]} grid={g} launches={n} avg_us={us:.2f}")
        print(f"      {cn:28s} {v:16.1f}")
PY
rm -rf $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2 $OUT/tcc
grep -A12 'grid=524288' $OUT/summary.txt | head -120

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
run() { name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o p -- python3 $ROOTDIR/tools/tune.py child > $OUT/$name.log 2>&1; }
if [ "${PMC_SET:-base}" = "mem" ]; then
run sq1 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VMEM_WR_TA_DATA_FIFO_FULL
run sq2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
else
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
fi
python3 - "$OUT" <<'PY' > $OUT/summary.txt 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
for sub in ("fetch", "write", "sq1", "sq2", "tcc"):
    f = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    if not f:
        lg = os.path.join(out, sub + ".log")
        print(sub, "no db;", open(lg).read()[-300:] if os.path.exists(lg) else "not run"); continue
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

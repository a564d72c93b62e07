# SQ counters of the finest-level fused sweep (default leg; BENCH_ARGS="--cells 288 --levels 6" for other sizes): bash tools/pmc_sweep.sh [extra env assignments...]
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | cut -c1-3000 > $ROOTDIR/gpurun_out/sq_counters.txt
G1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU"
G3="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  timeout -k 5 500 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc$i -o b -- python3 $ROOTDIR/bench.py --no-cpu-baseline --no-varcoef --no-weak-ref --steps 3 --warmup 1 $BENCH_ARGS > /tmp/pmc$i.log 2>&1 < /dev/null
  python3 - <<PY
import glob, sqlite3
f = glob.glob("/tmp/pmc$i/**/*.db", recursive=True)
if not f:
    print("no db for group $i"); print(open("/tmp/pmc$i.log").read()[-1500:])
else:
    con = sqlite3.connect(f[0])
    rows = con.execute("select kernel_name, grid_size_x, counter_name, avg(value), count(*) from counters_collection "
                       "where (kernel_name like '%sells_sweep%' or kernel_name like '%sells_rsweep%') group by kernel_name, grid_size_x, counter_name order by grid_size_x desc, kernel_name").fetchall()
    for r in rows:
        if r[1] >= 200000:
            print(f"{r[0][:60]:60s} grid {r[1]:8d} {r[2]:24s} {r[3]:16.0f}  n={r[4]}")
PY
done

# SQ counters of one kernel family in any command: bash tools/pmc_kernel.sh <kernel-name-substring> <min grid_x> -- <python script + args>
# (three --pmc passes; per kernel name and grid size: average counter values per launch)
PAT=$1; MING=$2; shift 3
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
G3="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1)); rm -rf /tmp/pmck$i
  timeout -k 5 900 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmck$i -o b -- python3 "$@" > /tmp/pmck$i.log 2>&1 < /dev/null
  python3 - "$PAT" "$MING" <<PY
import glob, sqlite3, sys
pat, ming = sys.argv[1], int(sys.argv[2])
f = glob.glob("/tmp/pmck$i/**/*.db", recursive=True)
if not f:
    print("no db for group $i"); print(open("/tmp/pmck$i.log").read()[-1500:])
else:
    con = sqlite3.connect(f[0])
    rows = con.execute("select kernel_name, grid_size_x, counter_name, avg(value), count(*) from counters_collection "
                       "where kernel_name like ? group by kernel_name, grid_size_x, counter_name order by grid_size_x desc, kernel_name", ("%" + pat + "%",)).fetchall()
    for r in rows:
        if r[1] >= ming:
            print(f"{r[0][:70]:70s} grid {r[1]:8d} {r[2]:26s} {r[3]:16.0f}  n={r[4]}")
PY
done

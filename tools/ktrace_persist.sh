# kernel-trace of the default leg with and without the one-launch smoothing pass
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for p in 0 1; do
  export GMG_PERSIST=$p
  rm -rf /tmp/kt$p
  timeout -k 5 500 rocprofv3 --kernel-trace --stats -d /tmp/kt$p -o b -- python3 $ROOTDIR/bench.py --no-cpu-baseline --no-varcoef --steps 5 > /tmp/kt$p.log 2>&1 < /dev/null
  echo "== GMG_PERSIST=$p"
  python3 $ROOTDIR/tools/kstats_db.py /tmp/kt$p 22 | tee $ROOTDIR/gpurun_out/persist${p}_kernel_stats.txt | cut -c1-170
done

#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [summarize flags] -- [bench.py args]
#   e.g.  bash profiles/run_profile.sh r04                 -- --legs default,generic,varcoef
#         bash profiles/run_profile.sh r04_288 --merge-latest -- --cells 288 --levels 6 --legs default,generic
#         bash profiles/run_profile.sh r04_config3_128 --merge-latest --order 2 --cells 128 --levels 5 -- --legs default,config3 --config3-cells 128
#         bash profiles/run_profile.sh r06_config5_1024 --merge-latest --order 3 --cells 1024 --levels 7 -- --legs default,config5 --config5-cells 1024 --config5-levels 7
# Produces gpurun_out/<tag>_kernel_stats.txt, _hbm_traffic.txt / .json (copy those into profiles/) and refreshes / extends
# profiles/traffic_latest.json.  Counters are collected in their own passes (--pmc + --kernel-trace only), as the MI355X guide prescribes.
TAG=${1:-r04}; shift
SFLAGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SFLAGS+=("$1"); shift; done
[ "$1" = "--" ] && shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 $*"
# (the printed line is a <= 4 KB extract since round 6: the sweep signatures summarize.py needs are in the details file of the trace pass)
export GMG_BENCH_DETAILS=$OUT/legs_trace.json
timeout -k 5 1500 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOTDIR/bench.py $ARGS > $OUT/trace.log 2>&1
export GMG_BENCH_DETAILS=$OUT/legs_pmc.json
timeout -k 5 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- python3 $ROOTDIR/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -k 5 1500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- python3 $ROOTDIR/bench.py $ARGS > $OUT/write.log 2>&1
unset GMG_BENCH_DETAILS
cd $ROOTDIR
# problem size of the run, for the records' labels (bench.py defaults: 128 cells, 4 levels)
CELLS=128; LEVELS=4; prev=""
for w in "$@"; do
  [ "$prev" = "--cells" ] && CELLS=$w
  [ "$prev" = "--levels" ] && LEVELS=$w
  prev=$w
done
HAVE_CELLS=0
for w in "${SFLAGS[@]}"; do [ "$w" = "--cells" ] && HAVE_CELLS=1; done
[ $HAVE_CELLS = 0 ] && SFLAGS+=(--cells $CELLS --levels $LEVELS)
python3 profiles/summarize.py $OUT gpurun_out/$TAG "${SFLAGS[@]}" --cmd "python3 bench.py $ARGS" > $OUT/summary.log 2>&1
grep -a '^{' $OUT/trace.log | tail -1 > gpurun_out/${TAG}_bench.json
cp $OUT/legs_trace.json gpurun_out/${TAG}_bench_legs.json 2>/dev/null
tail -3 $OUT/trace.log | cut -c1-400
tail -30 $OUT/summary.log
# keep the databases out of the merged gpurun_out (size limit): summaries only
rm -rf $OUT/trace $OUT/fetch $OUT/write

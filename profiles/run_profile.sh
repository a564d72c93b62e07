#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [extra bench.py args]
# Produces gpurun_out/prof_<tag>/{trace,fetch,write}/... and the summaries gpurun_out/<tag>_*.txt|json
# (copy those into profiles/).  Counters are collected in their own passes (--pmc + --kernel-trace only).
TAG=${1:-r02}; shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-weak-ref $*"
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOTDIR/bench.py $ARGS > $OUT/trace.log 2>&1
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- python3 $ROOTDIR/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- python3 $ROOTDIR/bench.py $ARGS > $OUT/write.log 2>&1
cd $ROOTDIR
# problem size of the run, for the records' labels (bench.py defaults: 128 cells, 4 levels)
CELLS=128; LEVELS=4; prev=""
for w in "$@"; do
  [ "$prev" = "--cells" ] && CELLS=$w
  [ "$prev" = "--levels" ] && LEVELS=$w
  prev=$w
done
python3 profiles/summarize.py $OUT gpurun_out/$TAG --no-latest --cells $CELLS --levels $LEVELS --cmd "python3 bench.py $ARGS" > $OUT/summary.log 2>&1
tail -5 $OUT/trace.log | cut -c1-600
cat $OUT/summary.log | tail -30
# keep the databases out of the merged gpurun_out (size limit): summaries only
rm -rf $OUT/trace $OUT/fetch $OUT/write

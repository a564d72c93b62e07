#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag>
# Produces gpurun_out/prof_<tag>/{trace,fetch,write}/... ; summaries are then copied into profiles/.
TAG=${1:-r01}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOTDIR/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
find $OUT -type f | head -50

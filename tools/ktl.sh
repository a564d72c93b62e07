#!/bin/bash
# ordered kernel timeline (tools/ktimeline.py) of the last kernels of a bench.py run:  bash tools/ktl.sh <out.txt> [ENV=..] -- <bench args>
OUTF=$1; shift
ENVS=()
while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
for kv in "${ENVS[@]}"; do export "$kv"; done
timeout -k 5 600 rocprofv3 --kernel-trace -d /tmp/tl -o p -- python3 $R/bench.py "$@" > /tmp/tl.log 2>&1 < /dev/null
python3 $R/tools/ktimeline.py /tmp/tl -${KTL_N:-230} ${KTL_N:-230} > $R/$OUTF

#!/bin/bash
# Round 6: every profile the records of this round come from, one GPU lease (run from the repo root on the GPU box):
#   kernel-trace + FETCH_SIZE / WRITE_SIZE passes of the bench legs (profiles/run_profile.sh -> traffic_latest.json),
#   the ordered timeline of one CG iteration at 128^3, the full default bench.
set -x
bash profiles/run_profile.sh r06 -- --legs default,generic,varcoef
bash profiles/run_profile.sh r06_288 --merge-latest -- --cells 288 --levels 6 --legs default,generic
bash profiles/run_profile.sh r06_config3_256 --merge-latest --order 2 --cells 256 --levels 5 -- --legs default,config3
bash profiles/run_profile.sh r06_config5_1024 --merge-latest --order 3 --cells 1024 --levels 7 -- --legs default,config5
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && timeout -k 5 600 rocprofv3 --kernel-trace -d /tmp/tl -o p -- python3 $R/bench.py --legs default --steps 20 > /tmp/tl.log 2>&1 < /dev/null; python3 $R/tools/iteration_timeline.py /tmp/tl > $R/gpurun_out/r06_timeline_128.txt 2>&1 )
cp gpurun_out/traffic_latest.json profiles/traffic_latest.json     # (summarize.py keeps it next to its outputs; bench.py reads profiles/)
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
echo "bench rc=$?"; wc -c gpurun_out/r06_bench_line.json

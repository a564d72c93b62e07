R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmc256; cd /tmp; export TMPDIR=/tmp
export TUNE_VARIANTS='[{"GMG_XCD_REMAP":0},{"GMG_XCD_REMAP":1}]'
timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc256/fetch -o p -- python3 $R/tools/tune256.py child > $R/gpurun_out/pmc256/log.txt 2>&1
tail -3 $R/gpurun_out/pmc256/log.txt

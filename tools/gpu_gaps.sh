#!/bin/bash
# kernel trace of the fp32 bench -> idle-gap report.  usage: tools/gpu_gaps.sh <tag> [bench args]
TAG=${1:-gaps}; shift; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> /dev/null
T=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_report.py $T 2 6 > $OUT/gaps.txt 2>&1; cat $OUT/gaps.txt

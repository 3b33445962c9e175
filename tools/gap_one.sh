#!/bin/bash
# kernel trace of one bench command + tools/gap_report.py (idle gaps, per-ms timeline): tools/gap_one.sh <tag> <name> <bench.py args...>
TAG=$1; NAME=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$NAME -o t -- python $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > /dev/null 2>&1
T=$(find $OUT/trace_$NAME -name "*kernel_trace.csv" | head -1)
(cd $R; python tools/gap_report.py $T 3 7 > $OUT/gaps_$NAME.txt 2>&1)
rm -rf $OUT/trace_$NAME

#!/bin/bash
# rocprofv3 --kernel-trace --stats of ONE bench command, summarised: tools/prof_one.sh <tag> <name> <bench.py args...>
TAG=$1; NAME=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$NAME -o b -- python $R/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > $OUT/bench_${NAME}_under_rocprof.json 2> /dev/null
S=$(find $OUT/prof_$NAME -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats_$NAME.csv && (cd $R; python tools/stats_summary.py $OUT/kernel_stats_$NAME.csv auto 80 > $OUT/summary_$NAME.txt)
rm -rf $OUT/prof_$NAME

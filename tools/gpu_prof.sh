#!/bin/bash
# rocprofv3 kernel-stats of the fp32 and bf16 bench (no tests): tools/gpu_prof.sh <tag>
TAG=${1:-prof}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in fp32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$d -o b -- python $R/bench.py --dtype $d --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > $OUT/bench_${d}_under_rocprof.json 2> /dev/null
  S=$(find $OUT/prof_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp $S $OUT/kernel_stats_$d.csv && (cd $R; python tools/stats_summary.py $OUT/kernel_stats_$d.csv auto 60 > $OUT/summary_$d.txt; head -12 $OUT/summary_$d.txt)
done
find $OUT -name "*.csv" -size +1M -delete

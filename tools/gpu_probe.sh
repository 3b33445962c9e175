#!/bin/bash
# targeted GPU visit: selected tests, a rocprofv3 kernel-stats pass of the bf16 bench, phase timing, PMC traffic of the fp32 bench
TAG=${1:-probe}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_ref_golden.py tests/test_gpu_model.py -m gpu -q -k "bf16 or golden or train_step or mlp" > $OUT/pytest_sel.txt 2>&1; tail -6 $OUT/pytest_sel.txt
timeout 300 python tools/phase_time.py > $OUT/phase_time.txt 2>&1; tail -14 $OUT/phase_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16 -o b -- python $R/bench.py --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_bf16_under_rocprof.json 2> /dev/null
cd $R
S=$(find $OUT/prof_bf16 -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats_bf16.csv && python tools/stats_summary.py $OUT/kernel_stats_bf16.csv 11 40 > $OUT/summary_bf16.txt && head -50 $OUT/summary_bf16.txt
find $OUT -name "*.csv" -size +1M -delete
if [ "$2" = "pmc" ]; then bash tools/pmc_bench.sh > $OUT/pmc.txt 2>&1; tail -16 $OUT/pmc.txt; fi

#!/bin/bash
# One GPU-box visit: the -m gpu suite (log kept), the default bench line, a rocprofv3 kernel-stats pass of the bench.
# usage: tools/gpu_round.sh <tag> [pytest-args...]
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
rm -f gpurun_out/parity_errors.jsonl
timeout 1500 python -m pytest tests -m gpu -q "$@" > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log; tail -c 1500 $OUT/bench.json
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.log; python -c "import json; d=json.load(open('$OUT/bench_bf16.json')); print('bf16:', d['value'], d['ms_per_step'], {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()})"
python -c "import json; d=json.load(open('$OUT/bench.json')); print('fp32:', d['value'], d['ms_per_step'], {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()})"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /dev/null
cd $R
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats.csv && python tools/stats_summary.py $OUT/kernel_stats.csv auto 45 > $OUT/summary.txt && head -12 $OUT/summary.txt
find $OUT/prof -name "*.csv" -size +1M -delete

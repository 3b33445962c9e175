#!/bin/bash
# Round-end evidence without the 13-minute full suite: bench lines (fp32 with the CPU baseline, bf16), rocprofv3 kernel stats of both,
# PMC HBM traffic + SQ issue breakdown of the fp32 step, smoke(), and the test files that cover what changed since the last full
# run.   usage: tools/gpu_final2.sh <tag> "<pytest files / -k expression>"
TAG=${1:-final2}; SEL=${2:-tests/test_gpu_bf16.py tests/test_gpu_dist.py}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.log
timeout 300 python bench.py --no-prefetch --no-cpu-baseline > $OUT/bench_noprefetch.json 2> /dev/null
python -c "
import json
for f in ('bench', 'bench_bf16', 'bench_noprefetch'):
    d = json.load(open('$OUT/' + f + '.json')); print(f, d['value'], d['ms_per_step'], {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()})"
bash tools/gpu_prof.sh $TAG > $OUT/prof.txt 2>&1; head -11 $OUT/prof.txt
bash tools/pmc_bench.sh > $OUT/pmc.txt 2>&1; cp gpurun_out/pmc_bench/summary.json $OUT/pmc_summary.json; tail -3 $OUT/pmc.txt | cut -c1-200
bash tools/pmc_sq.sh > $OUT/pmc_sq.txt 2>&1; head -14 $OUT/pmc_sq.txt
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
if [ "$SEL" != "none" ]; then timeout ${PYTEST_LIMIT:-400} python -m pytest $SEL -m gpu -q --timeout 300 --durations=12 > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; fi
tail -4 $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null

#!/bin/bash
# Round-end evidence in one GPU-box visit: full -m gpu suite (log kept), default bench lines (fp32 with the CPU baseline, bf16),
# rocprofv3 kernel stats of both, PMC traffic of the fp32 bench.   usage: tools/gpu_final.sh <tag>
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.log
python -c "
import json
for f in ('bench', 'bench_bf16'):
    d = json.load(open('$OUT/' + f + '.json')); print(f, d['value'], d['ms_per_step'], {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()})"
bash tools/gpu_prof.sh $TAG > $OUT/prof.txt 2>&1; head -11 $OUT/prof.txt
bash tools/pmc_bench.sh > $OUT/pmc.txt 2>&1; cp gpurun_out/pmc_bench/summary.json $OUT/pmc_summary.json; tail -4 $OUT/pmc.txt | cut -c1-200
timeout 1150 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -6 $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null

#!/bin/bash
# Round-end evidence on one MI355X: full -m gpu suite (log + parity errors kept), default bench line (fp32 headline + native-MFMA block +
# cfg3 block + cpu baseline), the cfg4 / cfg5 single-GPU lines, rocprofv3 kernel stats of the fp32 / bf16 / cfg4 / cfg5 bench, PMC traffic of
# the bench step (taken before the bench line, which quotes it).  usage: tools/gpu_final.sh <tag> [notests]
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
T0=$(date +%s)
if [ "$2" != "notests" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 1200 > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -6 $OUT/pytest_gpu.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
fi
echo "t=$(( $(date +%s) - T0 ))s"
# PMC traffic first: the bench line quotes it only when it was taken on exactly the kernel sources it times (bench.py _pmc_traffic)
PMC_TAG=${TAG}_pmc timeout 600 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; grep -E "_spconv_gmm" $OUT/pmc.log | cut -c1-200
cp gpurun_out/${TAG}_pmc/summary.json $OUT/pmc_traffic.json 2>/dev/null && cp $OUT/pmc_traffic.json profiles/round6_pmc_traffic.json
echo "t=$(( $(date +%s) - T0 ))s"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log || tail -5 $OUT/bench.log
python - <<PY
import json
d = json.load(open('$OUT/bench.json'))
print('fp32:', round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()}, d['config']['warmup_losses'])
c = d.get('cfg3')
if c: print('cfg3:', round(c['value'], 1), round(c['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in c['kernels'].items()})
n = d.get('fp32_native_mfma')
if n: print('native fp32 MFMAs:', round(n['value'], 1), round(n['ms_per_step'], 2))
print('cfg4 / cfg5 blocks of the default line:', {k: (round(d[k]['value'], 1), round(d[k]['ms_per_step'], 2)) for k in ('cfg4', 'cfg5') if k in d})
print('summary:', d.get('summary'))
print('cpu:', d.get('cpu_baseline', {}).get('value'), 'roofline:', {k: d['roofline'][k] for k in ('achieved', 'peak', 'frac', 'frac_of_dtype_peak', 'traffic') if k in d['roofline']})
PY
for c in cfg4 cfg5; do
timeout 400 python bench.py --config $c > $OUT/bench_$c.json 2>> $OUT/bench.log
python -c "
import json; d=json.load(open('$OUT/bench_$c.json')); print('$c', round(d['value'],1), round(d['ms_per_step'],2), {k:(round(v['ms_per_step'],2), v['frac_mfma'] and round(v['frac_mfma'],3)) for k,v in d['kernels'].items()}, d['config']['active_voxels_per_gpu'])"
done
echo "t=$(( $(date +%s) - T0 ))s"
cd /tmp && export TMPDIR=/tmp
for d in fp32 bf16 cfg4 cfg5; do
  ARGS="--dtype $d"; [ $d = cfg4 ] && ARGS="--config cfg4"; [ $d = cfg5 ] && ARGS="--config cfg5"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$d -o b -- python $R/bench.py $ARGS --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > $OUT/bench_${d}_under_rocprof.json 2> /dev/null
  S=$(find $OUT/prof_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp $S $OUT/kernel_stats_$d.csv && (cd $R; python tools/stats_summary.py $OUT/kernel_stats_$d.csv auto 60 > $OUT/summary_$d.txt; head -12 $OUT/summary_$d.txt)
done
find $OUT -name "*.csv" -size +1M -delete; rm -rf $OUT/prof_*
echo "t=$(( $(date +%s) - T0 ))s"

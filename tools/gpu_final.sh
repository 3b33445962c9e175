#!/bin/bash
# Round-end evidence on one MI355X: full -m gpu suite (log + parity errors kept), default bench line (fp32 headline + cfg3 block +
# cpu baseline), the same fp32 line with the native fp32 MFMA kernels (--fp32-math mfma), rocprofv3 kernel stats of the fp32 and bf16 bench, PMC traffic of the bench step.  usage: tools/gpu_final.sh <tag>
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -6 $OUT/pytest_gpu.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
echo "t=$(( $(date +%s) - T0 ))s"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log || tail -5 $OUT/bench.log
python - <<PY
import json
d = json.load(open('$OUT/bench.json'))
print('fp32:', round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()}, d['config']['warmup_losses'])
c = d.get('cfg3')
if c: print('cfg3:', round(c['value'], 1), round(c['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in c['kernels'].items()})
print('cpu:', d.get('cpu_baseline', {}).get('value'), 'traffic:', d['roofline']['traffic'])
PY
timeout 300 python bench.py --fp32-math mfma --no-cfg3 --no-cpu-baseline > $OUT/bench_fp32_mfma.json 2>> $OUT/bench.log
python -c "
import json; d = json.load(open('$OUT/bench_fp32_mfma.json')); print('fp32 with native fp32 MFMAs:', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()})"
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/gpu_prof.sh ${TAG}_prof > $OUT/prof.log 2>&1; grep -E "GPU busy|steps in" $OUT/prof.log
echo "t=$(( $(date +%s) - T0 ))s"
PMC_TAG=${TAG}_pmc bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; grep -E "_spconv_gmm|spconv_gmm_k<2" $OUT/pmc.log | cut -c1-200
echo "t=$(( $(date +%s) - T0 ))s"

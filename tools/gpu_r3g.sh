#!/bin/bash
# Round 3, GPU visit G: remaining tests on the current tree (NMS chunking, N>1 paths incl. the self-launching bench, cfg2 / cfg3 at full
# size with their final tolerances), then the evidence for profiles/: default bench line, kernel stats, PMC traffic.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3g; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_dist.py tests/test_gpu_full_size.py tests/test_gpu_model.py -m gpu -q --timeout 900 -k "not cfg4_joint and not large_dense" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt
tail -6 $OUT/pytest.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
echo "t=$(( $(date +%s) - T0 ))s"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log || tail -5 $OUT/bench.log
python - <<PY
import json
d = json.load(open('$OUT/bench.json'))
print('fp32:', round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()}, d['config']['warmup_losses'])
c = d.get('cfg3')
if c: print('cfg3:', round(c['value'], 1), round(c['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in c['kernels'].items()})
print('cpu:', d.get('cpu_baseline'))
PY
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/gpu_prof.sh r3g_prof > $OUT/prof.log 2>&1; tail -30 $OUT/prof.log | cut -c1-150
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; tail -16 $OUT/pmc.log | cut -c1-220
echo "t=$(( $(date +%s) - T0 ))s"

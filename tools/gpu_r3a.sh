#!/bin/bash
# Round 3, GPU visit A: validate the CPU-side work of the round (self-launching bench, cfg3 test + bench block, gradient parity on the
# product's activation pattern, SGD trajectory, weight-pack staleness fix, merged attention backward), measure, and take the counters
# on spconv_wgrad_k.  Parity asserts on gradients run SOFT (values are logged; tolerances get fixed from them).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3a; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
export U3D_PARITY_SOFT=1
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_gradients.py tests/test_gpu_dist.py tests/test_gpu_model.py tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "not large_dense_room" > $OUT/pytest_a.txt 2>&1; echo "exit $?" >> $OUT/pytest_a.txt
tail -6 $OUT/pytest_a.txt | cut -c1-300
echo "t=$(( $(date +%s) - T0 ))s"
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 800 -k "cfg3 or cfg2_full_size" > $OUT/pytest_b.txt 2>&1; echo "exit $?" >> $OUT/pytest_b.txt
tail -6 $OUT/pytest_b.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
echo "t=$(( $(date +%s) - T0 ))s"
unset U3D_PARITY_SOFT
# bench: default line (fp32 headline + cfg3 block + cpu baseline), then A/B of the kernel library against round 2's build
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log || tail -5 $OUT/bench.log
python - <<PY
import json
d = json.load(open('$OUT/bench.json'))
print('fp32:', round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v['ms_per_step'], 2), v['frac_mfma'] and round(v['frac_mfma'], 3)) for k, v in d['kernels'].items()}, d['config']['warmup_losses'])
c = d.get('cfg3')
if c: print('cfg3:', round(c['value'], 1), round(c['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in c['kernels'].items()}, c['config']['warmup_losses'])
print('cpu:', d.get('cpu_baseline'))
PY
echo "t=$(( $(date +%s) - T0 ))s"
U3D_LIB_PATH=$R/tools/bin/libu3d_r2.so timeout 300 python bench.py --no-cpu-baseline --no-cfg3 > $OUT/bench_r2lib.json 2> $OUT/bench_r2lib.log || tail -5 $OUT/bench_r2lib.log
python - <<PY
import json
d = json.load(open('$OUT/bench_r2lib.json'))
print('fp32 r2 lib:', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()}, d['config']['warmup_losses'])
PY

echo "t=$(( $(date +%s) - T0 ))s"
# counters on the sparse weight gradient (level 1 and 2)
timeout 400 bash tools/pmc_wgrad.sh 1 8 > $OUT/pmc_wgrad_l1.txt 2>&1; tail -8 $OUT/pmc_wgrad_l1.txt | cut -c1-400

echo "t=$(( $(date +%s) - T0 ))s"
# kernel stats of the fp32 bench step
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 > $OUT/bench_under_rocprof.json 2> /dev/null
cd $R
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats_fp32.csv && python tools/stats_summary.py $OUT/kernel_stats_fp32.csv auto 70 > $OUT/summary_fp32.txt && head -14 $OUT/summary_fp32.txt
find $OUT -name "*.csv" -size +1M -delete
echo "t=$(( $(date +%s) - T0 ))s"

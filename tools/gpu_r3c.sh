#!/bin/bash
# Round 3, GPU visit C: fused criterion (mixed / rotated), wgrad LDS pre-reduction, BN finish variants, eval chain; cfg4 full size.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3c; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_kernels.py tests/test_gpu_gradients.py tests/test_gpu_eval.py tests/test_gpu_bf16.py tests/test_gpu_postproc.py -m gpu -q --timeout 600 -k "not stress_1m" > $OUT/pytest_a.txt 2>&1; echo "exit $?" >> $OUT/pytest_a.txt
tail -8 $OUT/pytest_a.txt | cut -c1-300
echo "t=$(( $(date +%s) - T0 ))s"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py -m gpu -q --timeout 800 -k "end_to_end or cfg4_joint or prefetch or train_step" > $OUT/pytest_b.txt 2>&1; echo "exit $?" >> $OUT/pytest_b.txt
tail -8 $OUT/pytest_b.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
echo "t=$(( $(date +%s) - T0 ))s"
i=0
for v in "" "U3D_BN_FINISH=ticket" "U3D_WGRAD_TILES=512" "U3D_WGRAD_TILES=1024" "U3D_EPILOGUE_STATS=0"; do
  i=$((i+1))
  env $v timeout 200 python bench.py --no-cpu-baseline --no-cfg3 > $OUT/b$i.json 2> $OUT/b$i.log || tail -5 $OUT/b$i.log
  python -c "
import json
d = json.load(open('$OUT/b$i.json')); print('[$v]', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()}, d['config']['warmup_losses'][:2])"
done
echo "t=$(( $(date +%s) - T0 ))s"
cd /tmp && export TMPDIR=/tmp
for v in "" "U3D_BN_FINISH=ticket"; do
  tag=${v:+ticket}; tag=${tag:-default}
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 > /dev/null 2>&1
  S=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp $S $OUT/kernel_stats_$tag.csv && (cd $R; python tools/stats_summary.py $OUT/kernel_stats_$tag.csv auto 70 > $OUT/summary_$tag.txt; head -13 $OUT/summary_$tag.txt; grep -E "bn_|wgrad|crit" $OUT/summary_$tag.txt | cut -c1-120)
done
find $OUT -name "*.csv" -size +1M -delete
echo "t=$(( $(date +%s) - T0 ))s"

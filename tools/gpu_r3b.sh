#!/bin/bash
# Round 3, GPU visit B: kernel tests on the new sparse-conv epilogue (BN statistics), merged BN launches and the swizzled 128-byte
# accumulator rows; the tests visit A did not reach; A/B of the accumulator layout and of the epilogue statistics.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3b; mkdir -p $OUT; cd $R
rm -f gpurun_out/parity_errors.jsonl
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gradients.py tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_cabi.py -m gpu -q --timeout 600 -k "not large_dense_room and not stress_1m" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt
tail -12 $OUT/pytest.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
echo "t=$(( $(date +%s) - T0 ))s"
for lv in 1 2 3; do
  for lib in "" tools/bin/libu3d_ald40.so; do
    for r in "" 32; do
      echo "== level $lv lib=${lib:-default} R=${r:-plan}"
      U3D_LIB_PATH=${lib:+$R/$lib} U3D_GMM_R=$r timeout 120 python tools/prof_conv.py $lv 10 fwd 2>&1 | grep -E "spconv_gmm|level"
    done
  done
done > $OUT/prof_conv.txt 2>&1
cat $OUT/prof_conv.txt | cut -c1-200
echo "t=$(( $(date +%s) - T0 ))s"
i=0
for v in "" "U3D_LIB_PATH=$R/tools/bin/libu3d_ald40.so" "U3D_EPILOGUE_STATS=0" "U3D_GMM_R=32"; do
  i=$((i+1))
  env $v timeout 200 python bench.py --no-cpu-baseline --no-cfg3 > $OUT/b$i.json 2> $OUT/b$i.log || tail -5 $OUT/b$i.log
  python -c "
import json
d = json.load(open('$OUT/b$i.json')); print('[$v]', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()}, d['config']['warmup_losses'][:2])"
done
echo "t=$(( $(date +%s) - T0 ))s"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cfg3 > $OUT/bench_under_rocprof.json 2> /dev/null
cd $R
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats_fp32.csv && python tools/stats_summary.py $OUT/kernel_stats_fp32.csv auto 70 > $OUT/summary_fp32.txt && head -14 $OUT/summary_fp32.txt && grep "bn_" $OUT/summary_fp32.txt | cut -c1-110
find $OUT -name "*.csv" -size +1M -delete
echo "t=$(( $(date +%s) - T0 ))s"

#!/bin/bash
# Round 3, GPU visit D: timing ablations of spconv_gmm_k / spconv_wgrad_k (what bounds them), tests of the rulebook / GEMM / NMS changes,
# bench with the current defaults.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3d; mkdir -p $OUT; cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_postproc.py tests/test_gpu_model.py tests/test_gpu_full_size.py -m gpu -q --timeout 600 -k "rulebook or dense_linear or mlp_fused or ln_linear or nms or decoder or cfg2_rulebooks or voxelize or conv" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt | cut -c1-300
echo "t=$(( $(date +%s) - T0 ))s"
for lv in 1 2; do
  for lib in "" abl1 abl2 abl3; do
    echo "== gmm level $lv lib=${lib:-default}"
    U3D_LIB_PATH=${lib:+$R/tools/bin/libu3d_$lib.so} timeout 120 python tools/prof_conv.py $lv 10 fwd 2>&1 | grep -E "spconv_gmm"
  done
  for lib in "" wabl1 wabl3; do
    echo "== wgrad level $lv lib=${lib:-default}"
    U3D_LIB_PATH=${lib:+$R/tools/bin/libu3d_$lib.so} timeout 120 python tools/prof_conv.py $lv 10 wgrad 2>&1 | grep -E "spconv_wgrad"
  done
done > $OUT/ablations.txt 2>&1
cat $OUT/ablations.txt | cut -c1-160
echo "t=$(( $(date +%s) - T0 ))s"
timeout 200 python bench.py --no-cpu-baseline --no-cfg3 > $OUT/b1.json 2> $OUT/b1.log || tail -5 $OUT/b1.log
python -c "
import json
d = json.load(open('$OUT/b1.json')); print(round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()}, d['config']['warmup_losses'][:2])"
echo "t=$(( $(date +%s) - T0 ))s"

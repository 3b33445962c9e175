#!/bin/bash
# Round 3, GPU visit F: direct per-lane index loads in spconv_gmm_k, 64x64 NT GEMM tiles + tile cost model -- tests, timing, bench A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3f; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_model.py -m gpu -q --timeout 300 -k "conv or batchnorm or dense_linear or mlp_fused or ln_linear or decoder or linear" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt | cut -c1-200
for lv in 1 2 3 4; do timeout 120 python tools/prof_conv.py $lv 10 fwd 2>&1 | grep -E "spconv_gmm"; done | tee $OUT/prof_conv.txt
i=0
for v in "" "U3D_NT_TILE=2" "U3D_NT_TILE=3"; do
  i=$((i+1))
  env $v timeout 200 python bench.py --no-cpu-baseline --no-cfg3 > $OUT/b$i.json 2> $OUT/b$i.log || tail -5 $OUT/b$i.log
  python -c "
import json
d = json.load(open('$OUT/b$i.json')); print('[$v]', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()}, d['config']['warmup_losses'][:2], round(d['roofline']['frac'], 4))"
done

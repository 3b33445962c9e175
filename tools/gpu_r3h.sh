#!/bin/bash
# Round 3, GPU visit H: whole-row / LDS-transposition weight-gradient kernel for narrow rows -- tests, A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3h; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -m gpu -q --timeout 300 -k "conv" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt | cut -c1-200
for lv in 1; do for v in 1 0; do echo "== wgrad level $lv rows=$v"; U3D_WGRAD_ROWS=$v timeout 120 python tools/prof_conv.py $lv 10 wgrad 2>&1 | grep -E "spconv_wgrad"; done; done | tee $OUT/prof.txt
i=0
for v in "" "U3D_WGRAD_ROWS=0"; do
  i=$((i+1))
  env $v timeout 200 python bench.py --no-cpu-baseline --no-cfg3 > $OUT/b$i.json 2> $OUT/b$i.log || tail -5 $OUT/b$i.log
  python -c "
import json
d = json.load(open('$OUT/b$i.json')); print('[$v]', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()}, d['config']['warmup_losses'][:2])"
done

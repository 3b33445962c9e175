#!/bin/bash
# Round 3, GPU visit J: hashed voxel index (forced on an ordinary batch; automatic on a 15 000^2-cell extent) + the bitmap path again.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3j; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout 600 -k "hashed or large_extent or rulebook or voxelize or index or empty_and_single or backbone_features" > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt | cut -c1-300
U3D_INDEX=hash timeout 300 python bench.py --no-cpu-baseline --no-cfg3 --steps 3 --warmup 2 > $OUT/b_hash.json 2> $OUT/b_hash.log || tail -5 $OUT/b_hash.log
python -c "
import json
d = json.load(open('$OUT/b_hash.json')); print('[U3D_INDEX=hash]', round(d['value'], 1), round(d['ms_per_step'], 2), d['config']['warmup_losses'])"

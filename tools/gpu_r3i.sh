#!/bin/bash
# Round 3, GPU visit I: HBM traffic of spconv_gmm_k by PMC for three builds (current; previous index scheme with the swizzled
# accumulator; previous index scheme with the padded accumulator of rounds 1-2) -- which change moved 66 -> 77 MB per launch.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "" oldidx oldidx40; do
  echo "== build ${v:-current}"
  PMC_TAG=pmc_${v:-current} U3D_LIB_PATH=${v:+$R/tools/bin/libu3d_$v.so} bash tools/pmc_bench.sh 2>&1 | grep -E "spconv_gmm_k|_spconv_gmm" | cut -c1-200
  python -c "
import json; d = json.load(open('gpurun_out/pmc_${v:-current}/summary.json')); print('ALL gmm:', d['_spconv_gmm_k_all'])"
done

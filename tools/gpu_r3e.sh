#!/bin/bash
# Round 3, GPU visit E: what the non-MFMA half of spconv_gmm_k consists of (cumulative ablations), decoder test re-run.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3e; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 300 -k "decoder" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt | cut -c1-200
for lv in 1 2; do
  for a in "" 1 3 9 11 27 59 63; do
    echo "== gmm level $lv ablation mask ${a:-0}"
    U3D_LIB_PATH=${a:+$R/tools/bin/libu3d_abl$a.so} timeout 120 python tools/prof_conv.py $lv 10 fwd 2>&1 | grep -E "spconv_gmm"
  done
done > $OUT/ablations.txt 2>&1
cat $OUT/ablations.txt | cut -c1-160

#!/bin/bash
# TN GEMM microbench (two workgroup targets) + the dense-layer parity tests.  usage: tools/gpu_tn.sh <tag>
TAG=${1:-tn}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for w in 384 768; do echo "== U3D_TN_WGS=$w"; U3D_TN_WGS=$w timeout 120 python tools/prof_tn.py 2>&1 | grep -v bf16; done > $OUT/tn.txt 2>&1
cat $OUT/tn.txt
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py -q -x --timeout 200 -k "linear or mlp or dense or ln_linear" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt

timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" 2>&1 | tail -3
for lv in 1 2 3 4 5; do for m in mfma bf16x3; do U3D_WGRAD_X3_MIN=0 U3D_FP32_MATH=$m python tools/prof_wgrad.py 8 $lv fp32 | sed "s/^/$m /" | cut -c1-60,120-220; done; done

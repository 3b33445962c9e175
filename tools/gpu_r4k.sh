mkdir -p gpurun_out/r4k; rm -f gpurun_out/parity_errors.jsonl
timeout 900 python -m pytest tests/test_gpu_bf16.py -q --timeout 600 > gpurun_out/r4k/pytest_bf16.txt 2>&1; tail -5 gpurun_out/r4k/pytest_bf16.txt | cut -c1-250
U3D_PARITY_SOFT=1 timeout 1200 python -m pytest tests/test_gpu_full_size.py -q --timeout 1000 -k "cfg3" -s > gpurun_out/r4k/pytest_cfg3.txt 2>&1; tail -5 gpurun_out/r4k/pytest_cfg3.txt | cut -c1-400
grep -o '"grad_[a-z_]*": [0-9.e-]*' gpurun_out/r4k/pytest_cfg3.txt
cp gpurun_out/parity_errors.jsonl gpurun_out/r4k/

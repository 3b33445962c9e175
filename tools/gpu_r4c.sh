mkdir -p gpurun_out/r4c; rm -f gpurun_out/parity_errors.jsonl
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/r4c/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r4c/pytest_gpu.txt
tail -25 gpurun_out/r4c/pytest_gpu.txt | cut -c1-300
cp gpurun_out/parity_errors.jsonl gpurun_out/r4c/ 2>/dev/null

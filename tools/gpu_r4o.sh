mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests/test_gpu_bf16.py -q --timeout 600 > gpurun_out/r4o/pytest_bf16.txt 2>&1; tail -4 gpurun_out/r4o/pytest_bf16.txt | cut -c1-250
for i in 1 2; do timeout 400 python bench.py --no-cpu-baseline --no-mfma-line --dtype bf16 2>/dev/null | python -c "
import json,sys; c=json.loads(sys.stdin.read()); print('cfg3', round(c['value'],1), round(c['ms_per_step'],2), {k:(round(v['ms_per_step'],2)) for k,v in c['kernels'].items()})"; done

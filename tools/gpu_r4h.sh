mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_bf16.py -q --timeout 600 > gpurun_out/r4h/pytest_bf16.txt 2>&1; tail -25 gpurun_out/r4h/pytest_bf16.txt | cut -c1-250
timeout 400 python bench.py --no-cpu-baseline --no-mfma-line --dtype bf16 > gpurun_out/r4h/bench_bf16.json 2> gpurun_out/r4h/bench.log; python -c "
import json; c=json.load(open('gpurun_out/r4h/bench_bf16.json')); print('cfg3 rows', c['value'], c['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in c['kernels'].items()})"
U3D_BF16_ROWS=0 timeout 400 python bench.py --no-cpu-baseline --no-mfma-line --dtype bf16 > gpurun_out/r4h/bench_bf16_norows.json 2>> gpurun_out/r4h/bench.log; python -c "
import json; c=json.load(open('gpurun_out/r4h/bench_bf16_norows.json')); print('cfg3 fp32 rows', c['value'], c['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in c['kernels'].items()})"

mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_gpu_bf16.py -q --timeout 600 -k "rows or shadow" > gpurun_out/r4i/pytest_bf16.txt 2>&1; tail -25 gpurun_out/r4i/pytest_bf16.txt | cut -c1-250
timeout 300 python tools/prof_wgrad_rows.py 5 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i/prof_wgrad_rows.txt
timeout 400 python bench.py --no-cpu-baseline --no-mfma-line --dtype bf16 > gpurun_out/r4i/bench_bf16.json 2> gpurun_out/r4i/bench.log; python -c "
import json; c=json.load(open('gpurun_out/r4i/bench_bf16.json')); print('cfg3 rows', c['value'], c['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in c['kernels'].items()})"

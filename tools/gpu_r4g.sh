mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "conv" --timeout 600 > gpurun_out/r4g/pytest_conv.txt 2>&1; tail -4 gpurun_out/r4g/pytest_conv.txt | cut -c1-250
timeout 400 python bench.py --no-cpu-baseline --no-mfma-line > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.log; python -c "
import json; d=json.load(open('gpurun_out/r4g/bench.json')); print(d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in d['kernels'].items()}); c=d.get('cfg3'); print('cfg3', c and (c['value'], c['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in c['kernels'].items()}))"

mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" --timeout 600 > gpurun_out/r4a/pytest_conv.txt 2>&1; tail -15 gpurun_out/r4a/pytest_conv.txt | cut -c1-250
timeout 300 python tools/prof_gmm.py 10 x3 > gpurun_out/r4a/prof_gmm_x3.txt 2>&1; cat gpurun_out/r4a/prof_gmm_x3.txt | cut -c1-250
timeout 300 python tools/prof_gmm.py 10 bf16 > gpurun_out/r4a/prof_gmm_bf16.txt 2>&1; cat gpurun_out/r4a/prof_gmm_bf16.txt | cut -c1-250
timeout 400 python bench.py --no-cpu-baseline --no-mfma-line > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.log; python -c "
import json; d=json.load(open('gpurun_out/r4a/bench.json')); print(d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in d['kernels'].items()}); c=d.get('cfg3'); print('cfg3', c and (c['value'], c['ms_per_step']))"
U3D_GMM_WG=0 timeout 300 python bench.py --no-cpu-baseline --no-mfma-line --no-cfg3 > gpurun_out/r4a/bench_wave.json 2>> gpurun_out/r4a/bench.log; python -c "
import json; d=json.load(open('gpurun_out/r4a/bench_wave.json')); print('wave-tile:', d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in d['kernels'].items()})"

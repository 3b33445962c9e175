mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --timeout 600 -k "conv" > gpurun_out/r4l/pytest_conv.txt 2>&1; tail -6 gpurun_out/r4l/pytest_conv.txt | cut -c1-250
for v in 0 1; do for lv in 1 2; do echo "rows=$v level $lv: $(U3D_WGRAD_ROWS=$v timeout 120 python tools/prof_conv.py $lv 5 wgrad 2>&1 | grep -E 'wgrad \(')"; done; done
timeout 400 python bench.py --no-cpu-baseline --no-mfma-line --no-cfg3 > gpurun_out/r4l/bench.json 2> gpurun_out/r4l/bench.log; python -c "
import json; d=json.load(open('gpurun_out/r4l/bench.json')); print(d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],2)) for k,v in d['kernels'].items()})"

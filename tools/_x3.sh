mkdir -p gpurun_out/x3
python tools/bias_probe.py 2>&1 | tail -4
rm -f gpurun_out/parity_errors.jsonl
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_bf16.py -m gpu -q -k "cfg2 or conv or attention or linear or mlp" 2>&1 | tail -3
cp gpurun_out/parity_errors.jsonl gpurun_out/x3/parity_lo.jsonl
U3D_FP32_MATH=bf16x3 timeout 300 python bench.py --no-cpu-baseline --no-cfg3 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v.get('ms_per_step',0),2) for k,v in d.get('kernels',{}).items()}, d['config'].get('warmup_losses'))"

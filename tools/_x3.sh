timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" -s 2>&1 | grep -v "^$" | tail -12
for m in mfma bf16x3; do
  U3D_FP32_MATH=$m timeout 300 python bench.py --no-cpu-baseline --no-cfg3 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], {k:round(v.get('ms_per_step',0),2) for k,v in d.get('kernels',{}).items()}, d['config'].get('warmup_losses'))"
done

PROF_KINDS=wave timeout 120 python tools/prof_gmm.py 10 bf16rows 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py -q --timeout 600 -k "rows or conv" 2>&1 | tail -3
for i in 1 2; do timeout 400 python bench.py --no-cpu-baseline --no-mfma-line --dtype bf16 2>/dev/null | python -c "
import json,sys; c=json.loads(sys.stdin.read()); print('cfg3', round(c['value'],1), round(c['ms_per_step'],2), {k:(round(v['ms_per_step'],2)) for k,v in c['kernels'].items()})"; done

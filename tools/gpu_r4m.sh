for v in 0 1; do
U3D_WGRAD_SIDE_STREAM=$v timeout 300 python bench.py --no-cpu-baseline --no-mfma-line --no-cfg3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32 side=$v', round(d['value'],1), round(d['ms_per_step'],2))"
U3D_WGRAD_SIDE_STREAM=$v timeout 300 python bench.py --no-cpu-baseline --no-mfma-line --dtype bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 side=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done

for env in "" "U3D_EPILOGUE_STATS=1" "U3D_CONV_TS=1" "U3D_CONV_RS=1" ""; do
  echo "== ${env:-default}"
  env $env python bench.py --steps 20 --warmup 3 --no-cfg3 --no-mfma-line --no-extra-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2),'scenes/s', round(d['ms_per_step'],3),'ms/step; conv_gmm', round(d['kernels']['conv_gmm']['ms_per_step'],3), 'ms; loss', d['config']['loss'])"
done
echo "== cfg3 default / U3D_CONV_RS_BF16=1"
for env in "" "U3D_CONV_RS_BF16=1"; do env $env python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cfg3 --no-mfma-line --no-extra-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2),'scenes/s', round(d['ms_per_step'],3),'ms/step; conv_gmm', round(d['kernels']['conv_gmm']['ms_per_step'],3), 'ms; loss', d['config']['loss'])"; done

mkdir -p gpurun_out/r4r
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ref_golden.py tests/test_gpu_full_size.py -q --timeout 1200 -k "joint or cfg4 or mixed or golden or heading" > gpurun_out/r4r/pytest.txt 2>&1; tail -6 gpurun_out/r4r/pytest.txt | cut -c1-300
timeout 400 python bench.py --config cfg4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg4', round(d['value'],1), round(d['ms_per_step'],2))"

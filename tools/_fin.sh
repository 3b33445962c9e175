R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3fin; mkdir -p $OUT; cd $R
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.log || tail -5 $OUT/bench.log
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac'], d['cfg3']['value'], d['cpu_baseline']['value'], {k:(round(v['ms_per_step'],2), v['frac_mfma'] and round(v['frac_mfma'],3)) for k,v in d['kernels'].items()})"
timeout 300 python bench.py --fp32-math mfma --no-cfg3 --no-cpu-baseline > $OUT/bench_fp32_mfma.json 2>> $OUT/bench.log
python -c "
import json; d=json.load(open('$OUT/bench_fp32_mfma.json')); print('mfma', d['value'], d['ms_per_step'])"
bash tools/gpu_prof.sh r3fin_prof > $OUT/prof.log 2>&1; grep -E "GPU busy|steps in" $OUT/prof.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

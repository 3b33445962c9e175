#!/bin/bash
# quick GPU check: selected tests, then bench lines under env / flag variants.  usage: tools/gpu_quick.sh <tag> "<pytest -k expr>" [bench-arg-string ...]
TAG=$1; K=$2; shift 2; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
if [ -n "$K" ]; then timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "$K" > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt | cut -c1-220; fi
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 200 python bench.py --no-cpu-baseline $a > $OUT/b$i.json 2> $OUT/b$i.log || tail -5 $OUT/b$i.log
  python -c "
import json
d = json.load(open('$OUT/b$i.json')); print('[$a]', round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()})"
done

#!/bin/bash
# bench.py under a list of environment settings ("VAR=val" words; "-" = none).   usage: tools/gpu_envsweep.sh <tag> <setting> [<setting> ...]
TAG=$1; shift; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
i=0
for s in "$@"; do
  i=$((i+1)); e=""; [ "$s" != "-" ] && e="$s"
  env $e timeout 200 python bench.py --no-cpu-baseline $BENCH_ARGS > $OUT/b$i.json 2> $OUT/b$i.log
  python -c "
import json
d = json.load(open('$OUT/b$i.json')); print('$s', round(d['value'], 1), round(d['ms_per_step'], 2), d['config']['loss'], {k: round(v['ms_per_step'], 2) for k, v in d['kernels'].items()})"
done

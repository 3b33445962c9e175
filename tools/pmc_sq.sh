#!/bin/bash
# SQ issue/wait breakdown per kernel of the bench step (one PMC pass, kernel-trace only).  usage: tools/pmc_sq.sh [bench args]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/s -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs --no-prefetch "$@" > /dev/null 2> $OUT/err.txt
cd $R
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(f'{out}/s/*counter_collection.csv')
if not f:
    print('no counter file'); print(open(f'{out}/err.txt').read()[-2000:]); sys.exit(0)
res = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'u3d::' not in k: continue
    res[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
rows = sorted(res.items(), key=lambda kv: -kv[1]['SQ_BUSY_CYCLES'])[:28]
print(f'{"kernel":48s} {"n":>4s} {"wait_any":>8s} {"wait_inst":>9s} {"active":>7s} {"valu":>6s} {"vmem":>6s} {"mfma_busy/busy":>14s}')
lines = []
for k, v in rows:
    wc = max(v['SQ_WAVE_CYCLES'], 1)
    l = (f'{k.replace("u3d::", "")[:48]:48s} {n[k]:4d} {v["SQ_WAIT_ANY"] / wc:8.2f} {v["SQ_WAIT_INST_ANY"] / wc:9.2f} {v["SQ_ACTIVE_INST_ANY"] / wc:7.2f} '
         f'{v["SQ_ACTIVE_INST_VALU"] / wc:6.2f} {v["SQ_ACTIVE_INST_VMEM"] / wc:6.2f} {v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CYCLES"], 1):14.3f}')
    lines.append(l); print(l)
open(f'{out}/summary.txt', 'w').write('\n'.join(lines) + '\n')
PY
rm -rf $OUT/s

#!/bin/bash
# SQ issue / wait / LDS breakdown per u3d kernel of an arbitrary command (two PMC passes, kernel-trace only).
# usage: tools/pmc_cmd.sh <cmd...>        e.g.  tools/pmc_cmd.sh python tools/prof_gemm.py 41000
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$(mktemp -d /tmp/pmccmd.XXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o s -- "$@" > /dev/null 2> $OUT/err_a.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/b -o s -- "$@" > /dev/null 2> $OUT/err_b.txt
cd $R
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for p in 'ab':
    f = glob.glob(f'{out}/{p}/*counter_collection.csv')
    if not f:
        print('no counter file for pass', p); print(open(f'{out}/err_{p}.txt').read()[-1500:]); continue
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'u3d::' not in k: continue
        res[k][p + r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES' and p == 'a': n[k] += 1
rows = sorted(res.items(), key=lambda kv: -kv[1]['aSQ_BUSY_CYCLES'])[:16]
print(f'{"kernel":44s} {"n":>4s} {"wait_any":>8s} {"wait_inst":>9s} {"active":>7s} {"valu":>6s} {"vmem":>6s} {"mfma/busy":>9s} | {"lds_act":>7s} {"wait_lds":>8s} {"conflict/idx":>12s} {"valu/wave":>9s} {"lds/wave":>8s}')
for k, v in rows:
    wc = max(v['aSQ_WAVE_CYCLES'], 1); wb = max(v['bSQ_WAVE_CYCLES'], 1); wv = max(v['bSQ_WAVES'], 1)
    print(f'{k.replace("u3d::", "")[:44]:44s} {n[k]:4d} {v["aSQ_WAIT_ANY"] / wc:8.2f} {v["aSQ_WAIT_INST_ANY"] / wc:9.2f} {v["aSQ_ACTIVE_INST_ANY"] / wc:7.2f} '
          f'{v["aSQ_ACTIVE_INST_VALU"] / wc:6.2f} {v["aSQ_ACTIVE_INST_VMEM"] / wc:6.2f} {v["aSQ_VALU_MFMA_BUSY_CYCLES"] / max(v["aSQ_BUSY_CYCLES"], 1):9.3f} | '
          f'{v["bSQ_ACTIVE_INST_LDS"] / wb:7.2f} {v["bSQ_WAIT_INST_LDS"] / wb:8.2f} {v["bSQ_LDS_BANK_CONFLICT"] / max(v["bSQ_LDS_IDX_ACTIVE"], 1):12.3f} {v["bSQ_INSTS_VALU"] / wv:9.0f} {v["bSQ_INSTS_LDS"] / wv:8.0f}')
PY
rm -rf $OUT

#!/bin/bash
# Where does a sparse-convolution launch spend its time?  TA / L1 / LDS / SQ counters of the forward launches of tools/prof_gmm.py
# (level 1-2 shapes, workgroup-tile kernel), several PMC passes, kernel-trace only.  usage: tools/pmc_gmm.sh [kinds]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$(mktemp -d /tmp/pmcgmm.XXXX); KINDS=${1:-workgroup}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; PROF_KINDS=$KINDS PROF_MAXLV=2 timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o s -- python $R/tools/prof_gmm.py 4 x3 > /dev/null 2> $OUT/err_$tag.txt; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
run b GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run c SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run d SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
cd $R
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in 'abcd':
    f = glob.glob(f'{out}/{p}/*counter_collection.csv')
    if not f:
        print('no counter file for pass', p, open(f'{out}/err_{p}.txt').read()[-600:]); continue
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'spconv_gmm' not in k: continue
        res[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES' and p == 'c': n[k] += 1
for k, v in res.items():
    print(k.replace('u3d::', ''), 'launches', n[k])
    for c, x in sorted(v.items()):
        print(f'    {c:44s} {x / max(n[k], 1):16.0f} per launch')
PY
rm -rf $OUT

#!/bin/bash
# HBM traffic of every kernel of the bench step from PMC counters (two separate passes, kernel-trace only):
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide coalesced streams
# (MI355X_MICROARCH.md, HBM section) -> doubled below; WRITE_SIZE is used as reported (uncalibrated).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${PMC_TAG:-pmc_bench}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > /dev/null 2>&1
if [ "${PMC_CFG3:-1}" != "0" ]; then     # the same two passes over the cfg3 command (bf16 operands, 16 scenes): cfg3.roofline.traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f3 -o f -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w3 -o w -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs > /dev/null 2>&1
fi
cd $R
python - "$OUT" <<'PY'
import csv, sys, glob, json, collections
out = sys.argv[1]
def collect(ftag, wtag):
    res = collections.defaultdict(lambda: dict(fetch_kb=0.0, write_kb=0.0, n=0))
    for tag, key in ((ftag, 'fetch_kb'), (wtag, 'write_kb')):
        f = glob.glob(f'{out}/{tag}/*counter_collection.csv')
        if not f: continue
        for r in csv.DictReader(open(f[0])):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'u3d::' not in k: continue
            res[k][key] += float(r['Counter_Value'])
            if tag == ftag: res[k]['n'] += 1
    summ = {}
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['fetch_kb']):
        n = max(v['n'], 1)
        summ[k] = dict(dispatches=n, fetch_MB_per_launch_corrected=2.0 * v['fetch_kb'] * 1024 / n / 1e6,
                       write_MB_per_launch=v['write_kb'] * 1024 / n / 1e6)
    gm = [v for k, v in res.items() if 'spconv_gmm' in k]          # wave-tile and workgroup-tile forward / input-gradient kernels
    n = sum(v['n'] for v in gm)
    if n:
        summ['_spconv_gmm_all'] = dict(dispatches=n, hbm_MB_per_launch=(2.0 * sum(v['fetch_kb'] for v in gm) + sum(v['write_kb'] for v in gm)) * 1024 / n / 1e6)
    wg = [v for k, v in res.items() if 'spconv_wgrad' in k]
    n = sum(v['n'] for v in wg)
    if n:
        summ['_spconv_wgrad_all'] = dict(dispatches=n, hbm_MB_per_launch=(2.0 * sum(v['fetch_kb'] for v in wg) + sum(v['write_kb'] for v in wg)) * 1024 / n / 1e6)
    return summ
summ = collect('f', 'w')
c3 = collect('f3', 'w3')
if c3:
    summ['cfg3'] = c3
import os
root = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, root)
import bench
summ['_meta'] = dict(git_head=bench.git_head(), csrc_sha16=bench.csrc_hashes(),        # every kernel source this pass covers; bench.py refuses the file on a mismatch
                     command='python bench.py [--dtype bf16 for the cfg3 block] --steps 2 --warmup 1 --no-cpu-baseline --no-cfg3 --no-mfma-line --no-extra-configs under rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes)',
                     correction='FETCH_SIZE x 2 (gfx950 wide coalesced streams, MI355X_MICROARCH.md), WRITE_SIZE as reported; KB -> bytes x 1024')
json.dump(summ, open(f'{out}/summary.json', 'w'), indent=1)
for k, v in list(summ.items())[:14]: print(k[:70], v if k != 'cfg3' else {kk: vv for kk, vv in v.items() if kk.startswith('_')})
PY
find $OUT -name "*.csv" -size +1M -delete

#!/bin/bash
# PMC passes for the conv kernel (separate passes: SQ counters, cache counters, HBM bytes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/prof_conv.py $1 5 ${2:-fwd} > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o p1 -- python $R/tools/prof_conv.py $1 3 ${2:-fwd} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/p2 -o p2 -- python $R/tools/prof_conv.py $1 3 ${2:-fwd} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -o p3 -- python $R/tools/prof_conv.py $1 3 ${2:-fwd} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -o p4 -- python $R/tools/prof_conv.py $1 3 ${2:-fwd} > /dev/null 2>&1
cd $R
for p in p1 p2 p3 p4; do
  f=$(find $OUT/$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'gmm' in k or 'wgrad' in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()}, 'dispatches', max(cnt[(k, c)] for c in d))
PY
done
cat $OUT/plain.txt
find $OUT -name "*.csv" -size +2M -delete

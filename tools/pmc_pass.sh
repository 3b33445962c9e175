#!/bin/bash
# One rocprofv3 PMC pass over a command; prints per-kernel counter means for kernels matching $FILTER (default gmm|wgrad).
# usage: tools/pmc_pass.sh "<COUNTER ...>" <cmd...>
R=$GRAFT_REPO_ROOT; C="$1"; shift
OUT=$(mktemp -d /tmp/pmc.XXXX); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then echo "no counter output for: $C"; tail -3 $OUT/log.txt; else python3 - "$f" "${FILTER:-gmm|wgrad}" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if re.search(sys.argv[2], k):
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()}, 'dispatches', max(cnt[(k, c)] for c in d))
PY
fi
rm -rf $OUT

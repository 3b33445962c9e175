"""Idle-gap report from a rocprofv3 --kernel-trace csv of bench.py: where does the GPU wait between kernels?
Steps are delimited by the voxeliser's first kernel (u3d::vox_mark_k, once per step); steps [lo, hi) are analysed (default: the
timed steps after 2 warm-up steps).  usage: python tools/gap_report.py <kernel_trace.csv> [lo hi]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 6)
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'vox_mark_k' in r[2]]
print(f'{len(rows)} dispatches, {len(marks)} steps')
sel = rows[marks[lo]:marks[hi]]
n_steps = hi - lo
span = sel[-1][1] - sel[0][0]
busy = 0
gaps = []
end = sel[0][0]
for s, e, name in sel:
    if s > end:
        gaps.append((s - end, prev, name))
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, name
print(f'span {span / n_steps / 1e6:.3f} ms/step, busy {busy / n_steps / 1e6:.3f}, idle {(span - busy) / n_steps / 1e6:.3f} in {len(gaps) / n_steps:.0f} gaps/step')
short = lambda n: n.replace('void ', '').replace('at::native::', '').replace('u3d::', '')[:70]
for thr in (2000, 5000, 20000, 100000):
    g = [x for x in gaps if x[0] >= thr]
    print(f'  gaps >= {thr / 1e3:.0f} us: {len(g) / n_steps:.1f}/step, {sum(x[0] for x in g) / n_steps / 1e6:.3f} ms/step')
by = defaultdict(lambda: [0, 0])
for d, p, n in gaps:
    k = (short(p), short(n))
    by[k][0] += d
    by[k][1] += 1
print('top (previous kernel -> next kernel) by idle time per step:')
for (p, n), (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'  {d / n_steps / 1e3:8.1f} us {c / n_steps:6.1f}x  {p}  ->  {n}')
# timeline of the first analysed step: per 1 ms bucket, busy time and the kernel that starts the bucket
s0 = rows[marks[lo]][0]
step = rows[marks[lo]:marks[lo + 1]]
nb = int((step[-1][1] - s0) / 1e6) + 1
busy_b = [0.0] * nb
first = [''] * nb
for s, e, name in step:
    b = int((s - s0) / 1e6)
    if not first[b]:
        first[b] = short(name)
    while s < e:
        b = int((s - s0) / 1e6)
        nxt = min(e, s0 + (b + 1) * 1000000)
        busy_b[b] += nxt - s
        s = nxt
print('timeline of one step (1 ms buckets): busy us, first kernel starting in the bucket')
for b in range(nb):
    print(f'  {b:3d} ms  {busy_b[b] / 1e3:7.1f} us  {first[b]}')

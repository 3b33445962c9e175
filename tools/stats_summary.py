"""Summarise a rocprofv3 --stats kernel_stats.csv: python tools/stats_summary.py file.csv [n_steps|auto] [n_rows]
The number of profiled steps is derived from the trace itself (auto, the default): the decoder runs attn_fwd_k once per layer and
step, so steps = calls(attn_fwd{,_bf16,_x3}_k) / 6 for the 6-layer ScanNet model (U3D_DECODER_LAYERS overrides 6) -- a hand-passed count
was wrong by one step in round 2 (warm-up + timed + instrumented steps all appear in the trace)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
import os
arg = sys.argv[2] if len(sys.argv) > 2 else 'auto'
if arg == 'auto':
    layers = int(os.environ.get('U3D_DECODER_LAYERS', '6'))
    calls = sum(int(r['Calls']) for r in rows if re.search(r'attn_fwd(_bf16|_x3)?_k', r['Name']))
    assert calls and calls % layers == 0, f'cannot derive the step count: {calls} attn_fwd launches for {layers} layers'
    steps = calls / layers
    print(f'steps in this trace: {steps:.0f} (= {calls} attn_fwd launches / {layers} decoder layers)')
else:
    steps = float(arg)
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'GPU busy {tot / steps / 1e6:.2f} ms/step, {sum(int(r["Calls"]) for r in rows) / steps:.0f} kernels/step')


def grp(n):
    if 'spconv_gmm' in n: return 'u3d conv fwd+dgrad (spconv_gmm_k)'
    if 'gmm_reduce' in n: return 'u3d conv offset-group reduce (gmm_reduce_k)'
    if 'spconv_wgrad' in n: return 'u3d conv wgrad'
    if 'attn_' in n: return 'u3d attention'
    if 'bn_' in n: return 'u3d batch norm'
    if 'gemm_' in n or 'layer_norm' in n or 'transpose_k' in n or 'gelu_' in n: return 'u3d dense GEMM / LayerNorm (decoder Linear, 1x1 conv)'
    if n.startswith('Cijk'): return 'hipBLASLt GEMM (decoder Linear, 1x1 conv)'
    if 'u3d::' in n: return 'u3d voxelise/rulebook/pool/other'
    if 'rocclr' in n: return 'copy/memset'
    return 'torch elementwise/reduce/optimizer'


g = collections.Counter(); c = collections.Counter()
for r in rows:
    g[grp(r['Name'])] += int(r['TotalDurationNs']); c[grp(r['Name'])] += int(r['Calls'])
for k, v in g.most_common():
    print(f'{k:44s} {v / steps / 1e6:8.2f} ms/step {c[k] / steps:8.0f} calls/step {100.0 * v / tot:5.1f}%')
print()
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r'at::native::', '', r['Name'])[:110]
    print(f"{int(r['TotalDurationNs']) / steps / 1e6:7.2f} ms/step {int(r['Calls']) / steps:7.0f} {float(r['AverageNs']) / 1e3:9.1f} us  {n}")

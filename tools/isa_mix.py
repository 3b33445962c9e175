#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -save-temps device assembly file:  tools/isa_mix.py <file.s> <name regex> [loop]
Prints, per matching kernel, the instruction counts of the whole body (or of its hottest loop: the innermost backward branch
target .. branch span with the most MFMAs), the register / LDS footprint and the occupancy it implies."""
import re
import sys
from collections import Counter


def kernels(text):
    for m in re.finditer(r'^(\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel', text, re.S | re.M):
        yield m.group(1), m.group(2)


def hottest_loop(body):
    lines = body.split('\n')
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r'^(\.LBB\w+):', l))}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\w+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = lines[labels[m.group(1)]:i + 1]
            n = sum('v_mfma' in x for x in span)
            if best is None or n > best[0] or (n == best[0] and len(span) < len(best[1])):
                best = (n, span)
    return '\n'.join(best[1]) if best else body


def main():
    text = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2])
    loop = len(sys.argv) > 3
    for name, body in kernels(text):
        if not pat.search(name):
            continue
        code = hottest_loop(body) if loop else body
        c = Counter(re.findall(r'^\s+([a-z][a-z_0-9]+)', code, re.M))
        groups = Counter()
        for k, v in c.items():
            g = ('mfma' if 'mfma' in k else 'ds' if k.startswith('ds_') else 'vmem' if k.startswith(('buffer_', 'global_', 'flat_'))
                 else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'other')
            groups[g] += v
        meta = dict(re.findall(r'\.amdhsa_(next_free_vgpr|accum_offset|group_segment_fixed_size)\s+(\d+)', body))
        print(name)
        print('  ', dict(groups), meta)
        print('  ', dict(c.most_common(18)))


main()

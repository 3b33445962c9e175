"""Per-kernel resource report from a hipcc -save-temps .s file: VGPRs, AGPR offset, SGPRs, LDS, scratch, spills, MFMA / VALU / LDS / VMEM counts.
usage: python tools/isa_report.py <file.s> [name filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
meta = {}
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', body) or [None, '?'])[1]
    meta[name] = dict(vgpr=g('next_free_vgpr'), acc=g('accum_offset'), sgpr=g('next_free_sgpr'), lds=g('group_segment_fixed_size'), scratch=g('private_segment_fixed_size'))
for name, md in meta.items():
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        dn = name
    if flt and flt not in dn:
        continue
    m = re.search(r'^' + re.escape(name) + r':(.*?)s_endpgm', s, re.S | re.M)
    body = m.group(1) if m else ''
    cnt = lambda pat: len(re.findall(pat, body))
    print(f"{dn[:100]}\n    vgpr {md['vgpr']} (agpr from {md['acc']}) sgpr {md['sgpr']} lds {md['lds']} scratch {md['scratch']} | mfma {cnt(r'v_mfma')} ds_read {cnt(r'ds_read|ds_load')} "
          f"ds_write {cnt(r'ds_write|ds_store')} buffer_load {cnt(r'buffer_load')} global {cnt(r'global_(load|store)')} barrier {cnt(r's_barrier')} "
          f"scratch_ops {cnt(r'scratch_')} v_accvgpr {cnt(r'v_accvgpr')} lines {body.count(chr(10))}")

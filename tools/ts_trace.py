"""Cycle stamps of the tile-stationary kernel's phases (variant built with -DU3D_TS_TRACE; U3D_LIB_PATH points at it).
usage: U3D_LIB_PATH=tools/bin/libu3d_ts_trace.so python tools/ts_trace.py [T] [H] [KG]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd import precision as P  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

T, H, KG = (int(a) for a in (sys.argv[1:4] + ['128', '128', '1'][len(sys.argv) - 1:]))
os.environ['U3D_TS_T'], os.environ['U3D_TS_H'], os.environ['U3D_TS_KG'] = str(T), str(H), str(KG)
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(8)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
rb = sparse.build_subm_rulebook(vb.coords, vb.index)
n = vb.coords.shape[0]
cs, cd = (int(v) for v in os.environ.get('TRACE_SHAPE', '32x32').split('x'))
x = torch.randn(n, cs, device=dev)
w = torch.randn(cd, 3, 3, 3, cs, device=dev) * 0.05
nt = (n + T - 1) // T
with P.fp32_math('bf16x3'), sparse.conv_ts(True):
    for _ in range(3):
        sparse.sparse_conv(x, w, rb)
    tr = torch.zeros(nt, 32, dtype=torch.int64, device=dev)
    os.environ['U3D_TS_TRACE_PTR'] = str(tr.data_ptr())
    sparse.sparse_conv(x, w, rb)
    torch.cuda.synchronize()
    os.environ.pop('U3D_TS_TRACE_PTR')
t = tr.cpu().numpy().astype(np.int64)
t0 = t[:, 0]
total = t[:, 31] - t0
print(f'T={T} H={H} KG={KG} {cs}->{cd}: {nt} tiles; kernel span {(t[:, 31].max() - t0.min())} cycles')
print(f'tile life: mean {total.mean():.0f} p50 {np.median(total):.0f} p90 {np.quantile(total, .9):.0f} cycles')
print(f'prologue (start -> first pass): mean {(t[:, 1] - t0).mean():.0f}')
for ps in range(5):
    ok = t[:, 5 + ps * 5] > 0
    if not ok.any():
        break
    a = t[ok]
    steps = a[:, 26 + ps]
    print(f'pass {ps}: {ok.sum()} tiles | wait prev barrier {(a[:, 2 + ps * 5] - a[:, 1 + ps * 5]).mean():.0f} | load phase {(a[:, 3 + ps * 5] - a[:, 2 + ps * 5]).mean():.0f}'
          f' | barrier {(a[:, 4 + ps * 5] - a[:, 3 + ps * 5]).mean():.0f} | steps {(a[:, 5 + ps * 5] - a[:, 4 + ps * 5]).mean():.0f} over {steps.mean():.1f} steps'
          f' = {((a[:, 5 + ps * 5] - a[:, 4 + ps * 5]) / np.maximum(steps, 1)).mean():.0f} cycles/step')
last = np.zeros(nt, np.int64)
for ps in range(5):
    last = np.maximum(last, t[:, 5 + ps * 5])
print(f'epilogue: mean {(t[:, 31] - last).mean():.0f}')

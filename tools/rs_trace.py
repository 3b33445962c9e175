"""Accumulated cycles per phase and wave of the register-stationary kernel (variant built with -DU3D_TS_TRACE).
usage: U3D_LIB_PATH=tools/bin/libu3d_ts_trace.so python tools/rs_trace.py [H=416]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd import precision as P  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 416
os.environ['U3D_RS_H'] = str(H)
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(8)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
rb = sparse.build_subm_rulebook(vb.coords, vb.index)
n = vb.coords.shape[0]
x = torch.randn(n, 32, device=dev)
w = torch.randn(32, 3, 3, 3, 32, device=dev) * 0.05
with P.fp32_math('bf16x3'), sparse.conv_rs(True):
    for _ in range(3):
        sparse.sparse_conv(x, w, rb)
    tr = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
    os.environ['U3D_TS_TRACE_PTR'] = str(tr.data_ptr())
    sparse.sparse_conv(x, w, rb)
    torch.cuda.synchronize()
    os.environ.pop('U3D_TS_TRACE_PTR')
t = tr.cpu().numpy().reshape(256, 4, 8).astype(np.float64)
names = ['weights+prologue', 'wait tile barrier', 'load phase (ids, rows, split, LDS writes)', 'wait load barrier', 'compute (28 items)', 'red write + barrier', 'sum + store']
tiles = (n + 63) // 64
print(f'H={H}: {tiles} tiles, {tiles / 256:.1f} per workgroup; cycles per TILE (mean over workgroups), by wave:')
per = t / (tiles / 256)
for i, nm in enumerate(names):
    print(f'  {nm:45s} ' + ' '.join(f'{per[:, wv, i].mean():8.0f}' for wv in range(4)) + (f'   (per launch: {t[:, 0, i].mean():.0f})' if i == 0 else ''))
print(f'  {"total":45s} ' + ' '.join(f'{per[:, wv, 1:].sum(1).mean():8.0f}' for wv in range(4)))

"""Micro-driver for profiling the sparse conv kernels at cfg2 geometry (8 scenes x 100k pts, 2 cm).
usage: python tools/prof_conv.py [level] [iters] [mode]   (mode: fwd | wgrad)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else 'fwd'
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(8)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
coords, shape, index = vb.coords, vb.spatial_shape, vb.index
for _ in range(level - 1):
    coords, shape, index, _rb = sparse.build_down_rulebook(coords, 8, shape)
rb = sparse.build_subm_rulebook(coords, index)
C = 32 * level
n = coords.shape[0]
x = torch.randn(n, C, device=dev)
w = torch.randn(C, 3, 3, 3, C, device=dev) * 0.05
pairs = rb.total_pairs
print(f'level {level}: n={n} C={C} pairs={pairs} ({pairs / n:.2f}/voxel) flops={2.0 * pairs * C * C / 1e9:.2f} GF', flush=True)
if mode == 'fwd':
    f = lambda: sparse.sparse_conv(x, w, rb)            # noqa: E731
else:
    xg = x.clone().requires_grad_(False)
    wg = w.clone().requires_grad_()
    go = torch.randn(n, C, device=dev)
    def f():
        y = sparse.sparse_conv(xg, wg, rb)
        y.backward(go)
from unidet3d_amd import _lib as L  # noqa: E402
for _ in range(2):
    f()
torch.cuda.synchronize()
for c in (0, 1):
    L.prof_enable(c, True)
t0 = time.perf_counter()
for _ in range(iters):
    f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
for c, nm in ((0, 'spconv_gmm (+reduce)'), (1, 'spconv_wgrad (+reduce)')):     # HIP-event time of the kernels alone
    ms, cnt, _w = L.prof_collect(c)
    L.prof_enable(c, False)
    if cnt:
        print(f'  {nm}: {ms / cnt * 1e3:.1f} us/launch over {cnt} launches -> {2.0 * pairs * C * C / (ms / cnt * 1e-3) / 1e12:.2f} TFLOP/s')
print(f'{mode}: {dt * 1e6:.1f} us/iter  -> {2.0 * pairs * C * C / dt / 1e12:.2f} TFLOP/s (fwd flops only)', flush=True)

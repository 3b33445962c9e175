"""Sparse weight-gradient kernel at cfg2 / cfg3 geometry: time per launch for a level, batch and operand dtype.
usage (GPU box): U3D_WGRAD_TILES=512 python tools/prof_wgrad.py [batch] [level] [fp32|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import _lib as L, ops, precision, sparse  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
dev = torch.device('cuda:0')
vb = ops.voxelize([torch.from_numpy(make_scene(i).points).to(dev) for i in range(B)], 0.02, 128)
coords, shape, index = vb.coords, vb.spatial_shape, vb.index
for _ in range(level - 1):
    coords, shape, index, _rb = sparse.build_down_rulebook(coords, B, shape)
rb = sparse.build_subm_rulebook(coords, index)
C = 32 * level
n = coords.shape[0]
x = torch.randn(n, C, device=dev); w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.05).requires_grad_()
go = torch.randn(n, C, device=dev)
pairs = rb.total_pairs
with precision.operands(mode):
    def f():
        w.grad = None
        sparse.sparse_conv(x, w, rb).backward(go)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    L.prof_enable(L.K_CONV_WGRAD, True)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ms, cnt, _ = L.prof_collect(L.K_CONV_WGRAD)
T = L.lib().u3d_spconv_wgrad_tile_rows(27, n, C, C)
print(f'B={B} level={level} {mode} tiles_cap={os.environ.get("U3D_WGRAD_TILES", "256")}: n={n} C={C} pairs={pairs} tile_rows={T} n_tiles={(n + T - 1) // T}  '
      f'wgrad+reduce {ms / cnt * 1e3:.1f} us/launch -> {2.0 * pairs * C * C / (ms / cnt * 1e-3) / 1e12:.1f} TFLOP/s')

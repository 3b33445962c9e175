"""A/B of the sparse-convolution weight gradient at cfg3 geometry (16 scenes x 100k pts): fp32-row kernels (what bf16-operand mode
uses without shadows) against u3d_spconv_wgrad_rows on bf16 shadows.  usage: python tools/prof_wgrad_rows.py [iters] [scenes]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import _lib as L  # noqa: E402
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd import precision as P  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(B)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
coords, shape, index = vb.coords, vb.spatial_shape, vb.index
levels = []
for lv in range(1, 3):
    levels.append((lv, coords, sparse.build_subm_rulebook(coords, index)))
    coords, shape, index, _rb = sparse.build_down_rulebook(coords, B, shape)
with P.operands('bf16'):
    for lv, c, rb in levels:
        n = c.shape[0]
        for cs, cd in ((32 * lv, 32 * lv),) + (((64, 32), (32, 64)) if lv == 1 else ()):
            x = torch.randn(n, cs, device=dev)
            go = torch.randn(n, cd, device=dev)
            w = (torch.randn(cd, 3, 3, 3, cs, device=dev) * 0.05).requires_grad_()
            row = f'level {lv} n={n} {cs}->{cd}:'
            for rows in (False, True):
                with P.bf16_rows_mode(rows):
                    def f():
                        xs, gs = x.clone(), go.clone()
                        if rows:
                            sparse.attach_shadow(xs, sparse.to_shadow(xs)); sparse.attach_shadow(gs, sparse.to_shadow(gs))
                        w.grad = None
                        sparse.sparse_conv(xs, w, rb).backward(gs)
                    for _ in range(2):
                        f()
                    torch.cuda.synchronize()
                    L.prof_enable(1, True)
                    for _ in range(iters):
                        f()
                    torch.cuda.synchronize()
                    ms, cnt, _w = L.prof_collect(1)
                    L.prof_enable(1, False)
                    row += f'  {"bf16 rows" if rows else "fp32 rows"} {ms / max(cnt, 1) * 1e3:7.1f} us'
            print(row, flush=True)

"""A/B of the two output-stationary sparse-convolution kernels (wave tiles / workgroup tiles) at cfg2 geometry
(8 scenes x 100k pts, 2 cm): forward launches of every (level, Cs -> Cd) shape of the U-Net, HIP-event time per launch.
usage: python tools/prof_gmm.py [iters] [operands: x3|bf16|bf16rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import _lib as L  # noqa: E402
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd import precision as P  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
operands = sys.argv[2] if len(sys.argv) > 2 else 'x3'
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(8)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
coords, shape, index = vb.coords, vb.spatial_shape, vb.index
levels = []
for lv in range(1, 6):
    levels.append((lv, coords, sparse.build_subm_rulebook(coords, index)))
    coords, shape, index, _rb = sparse.build_down_rulebook(coords, 8, shape)


def timed(f):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    L.prof_enable(0, True)
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    ms, cnt, _w = L.prof_collect(0)
    L.prof_enable(0, False)
    return ms / max(cnt, 1) * 1e3


KINDS = os.environ.get('PROF_KINDS', 'wave,workgroup').split(',')
MAXLV = int(os.environ.get('PROF_MAXLV', '5'))
MINLV = int(os.environ.get('PROF_MINLV', '1'))
total = {k: 0.0 for k in KINDS}
ctx = P.operands('bf16') if operands.startswith('bf16') else P.fp32_math('bf16x3')
with ctx:
    for lv, c, rb in levels[MINLV - 1:MAXLV]:
        n = c.shape[0]
        C = 32 * lv
        for cs, cd in ((C, C), (2 * C, C)) if lv < 5 else ((C, C),):
            x = torch.randn(n, cs, device=dev)
            if operands == 'bf16rows':          # the shadow a batch norm would have written: the conv gathers bf16 rows
                sparse.attach_shadow(x, sparse.to_shadow(x))
            w = torch.randn(cd, 3, 3, 3, cs, device=dev) * 0.05
            pairs = rb.total_pairs
            gf = 2.0 * pairs * cs * cd / 1e9
            row = f'level {lv} n={n:7d} {cs:3d}->{cd:3d} pairs/voxel {pairs / n:5.2f} {gf:6.2f} GF:'
            ys = {}
            for kind in KINDS:
                with P.conv_kernel('workgroup-all' if kind == 'workgroup' else kind):
                    us = timed(lambda: sparse.sparse_conv(x, w, rb))
                    ys[kind] = sparse.sparse_conv(x, w, rb)
                total[kind] += us
                row += f'  {kind} {us:7.1f} us ({gf / us * 1e-3:6.1f} TF/s)'
            if len(KINDS) == 2:
                row += '  equal' if torch.equal(ys['wave'], ys['workgroup']) else f'  DIFF {float((ys["wave"] - ys["workgroup"]).abs().max()):.3e}'
            print(row, flush=True)
print('sum of shapes:', {k: round(v, 1) for k, v in total.items()}, flush=True)

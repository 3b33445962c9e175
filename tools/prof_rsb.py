"""bf16-row register-stationary SubM kernel (spconv_rsb_k) vs the bf16-row workgroup-tile pair kernel (u3d_spconv_gmm_bf16a), cfg3 geometry
(16 scenes x 100k points).  usage: python tools/prof_rsb.py [n_scenes=16] [iters=10]   env PROF_H=256,320,448 PROF_WGS=0,512,1024"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import _lib as L  # noqa: E402
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd import precision as P  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(ns)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
coords, shape, index = vb.coords, vb.spatial_shape, vb.index
levels = []
for lv in range(1, 3):
    levels.append((lv, coords, sparse.build_subm_rulebook(coords, index)))
    coords, shape, index, _rb = sparse.build_down_rulebook(coords, ns, shape)


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


sparse._RS_MIN_ROWS = 1000
with P.operands('bf16'):
    for lv, c, rb in levels:
        n = c.shape[0]
        pairs = rb.total_pairs
        print(f'== level {lv}: n={n} pairs/row {pairs / n:.2f}', flush=True)
        C = 32 * lv
        for cs, cd in ((C, C), (2 * C, C), (C, 2 * C)):
            x = torch.randn(n, cs, device=dev)
            sparse.attach_shadow(x, sparse.to_shadow(x))
            w = torch.randn(cd, 3, 3, 3, cs, device=dev) * 0.05
            gf = 2.0 * pairs * cs * cd / 1e9
            with sparse.conv_rs_bf16(False):
                us0 = timed(lambda: sparse.sparse_conv(x, w, rb))
                y0 = sparse.sparse_conv(x, w, rb)
            print(f'   {cs:3d}->{cd:3d} {gf:6.2f} GF: pair kernel (bf16 rows) {us0:7.1f} us', flush=True)
            for H in [int(h) for h in os.environ.get('PROF_H', '320,448').split(',')]:
                for wgs in [int(v) for v in os.environ.get('PROF_WGS', '0').split(',')]:
                    os.environ['U3D_RSB_H'], os.environ['U3D_RS_WGS'] = str(H), str(wgs)
                    with sparse.conv_rs_bf16(True):
                        us1 = timed(lambda: sparse.sparse_conv(x, w, rb))
                        y1 = sparse.sparse_conv(x, w, rb)
                    err = float((y1 - y0).abs().max() / y0.abs().max())
                    print(f'      rsb H={H} wgs={wgs}: {us1:7.1f} us x{us0 / us1:4.2f}  max |diff| vs pair kernel {err:.2e}', flush=True)

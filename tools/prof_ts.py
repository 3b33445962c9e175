"""Tile-stationary SubM kernel (csrc/spconv_ts.hip): table check, accuracy against fp64 and A/B timing against the pair-list
workgroup-tile kernel at cfg2 geometry.
usage: python tools/prof_ts.py [n_scenes=8] [iters=10]   env PROF_PLANS="128x192,128x128,256x192" PROF_SHAPES="32x32,64x64" PROF_CHECK=1"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import _lib as L  # noqa: E402
from unidet3d_amd import ops, sparse  # noqa: E402
from unidet3d_amd import precision as P  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
scenes = [make_scene(i) for i in range(ns)]
vb = ops.voxelize([torch.from_numpy(s.points).to(dev) for s in scenes], 0.02, 128)
coords, shape, index = vb.coords, vb.spatial_shape, vb.index
levels = []
for lv in range(1, 4):
    levels.append((lv, coords, sparse.build_subm_rulebook(coords, index)))
    coords, shape, index, _rb = sparse.build_down_rulebook(coords, ns, shape)


def timed(f, cls=0):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    L.prof_enable(cls, True)
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    ms, cnt, _w = L.prof_collect(cls)
    L.prof_enable(cls, False)
    return ms / max(cnt, 1) * 1e3


def check_tables(rb, T, H):
    """halo / loc / pmask against numpy on the rulebook's own pair lists"""
    n = rb.n_out
    nbr = np.full((27, n), -1, np.int64)
    for k, (i, o) in enumerate(rb.lists()):
        nbr[k, o] = i
    nhalo, halo, loc, pmask = [t.cpu().numpy() for t in rb.halo(T, H)]
    loc = loc.view(np.uint16)
    pmask = pmask.view(np.uint32)
    nt = (n + T - 1) // T
    bad = 0
    for t in range(nt):
        blk = nbr[:, t * T:(t + 1) * T]
        u = np.unique(blk[blk >= 0])
        ok = nhalo[t] == len(u) and np.array_equal(halo[t, :len(u)], u)
        ref = np.full((27, T), 0xFFFF, np.int64)
        w = blk.shape[1]
        pos = np.searchsorted(u, blk)
        ref[:, :w] = np.where(blk >= 0, pos, 0xFFFF)
        ok = ok and np.array_equal(loc[t].astype(np.int64), ref)
        pm = np.zeros(pmask.shape[1], np.uint32)
        kk, rr = np.nonzero(blk >= 0)
        np.bitwise_or.at(pm, pos[kk, rr] // H, (np.uint32(1) << kk.astype(np.uint32)))
        ok = ok and np.array_equal(pmask[t], pm)
        bad += 0 if ok else 1
    return bad, nt, float(nhalo.sum()) / n


def ref64(x, w, rb, flip):
    """fp64 reference on the GPU from the pair lists: forward (flip = 0) or input gradient (flip = 1)"""
    cout, cin = w.shape[0], w.shape[-1]
    wk = w.reshape(cout, 27, cin).double()
    cnt = rb.counts.cpu().tolist()
    y = torch.zeros(rb.n_out, cin if flip else cout, dtype=torch.float64, device=x.device)
    xd = x.double()
    for k in range(27):
        i, o = rb.pair_in[k, :cnt[k]].long(), rb.pair_out[k, :cnt[k]].long()
        if flip:
            y.index_add_(0, i, xd[o] @ wk[:, k, :])
        else:
            y.index_add_(0, o, xd[i] @ wk[:, k, :].T)
    return y


plans = [tuple(int(v) for v in p.split('x')) for p in os.environ.get('PROF_PLANS', '128x192,128x128,128x256,256x128,256x192,256x256').split(',')]
shapes = [tuple(int(v) for v in p.split('x')) for p in os.environ.get('PROF_SHAPES', '32x32,64x32,32x64,64x64').split(',')]
check = os.environ.get('PROF_CHECK', '1') == '1'
maxlv = int(os.environ.get('PROF_MAXLV', '2'))
with P.fp32_math('bf16x3'):
    for lv, c, rb in levels[:maxlv]:
        n = c.shape[0]
        pairs = rb.total_pairs
        print(f'== level {lv}: n={n} pairs/row {pairs / n:.2f}', flush=True)
        if check:
            for T, H in plans[:2]:
                bad, nt, ratio = check_tables(rb, T, H)
                print(f'   tables T={T} H={H}: {bad} of {nt} tiles differ; halo rows / n = {ratio:.2f}', flush=True)
        for T, H in plans:
            us = timed(lambda: rb._halo.clear() or rb.halo(T, H), cls=L.K_RULEBOOK)
            print(f'   u3d_subm_halo T={T} H={H}: {us:7.1f} us', flush=True)
        for cs, cd in shapes:
            x = torch.randn(n, cs, device=dev)
            w = torch.randn(cd, 3, 3, 3, cs, device=dev) * 0.05
            go = torch.randn(n, cd, device=dev)
            gf = 2.0 * pairs * cs * cd / 1e9
            with sparse.conv_ts(False):
                us0 = timed(lambda: sparse.sparse_conv(x, w, rb))
                xg = x.clone().requires_grad_()
                y0 = sparse.sparse_conv(xg, w, rb)
                y0.backward(go)
                d0 = xg.grad
            row = f'   {cs:3d}->{cd:3d} {gf:6.2f} GF: pairs kernel {us0:7.1f} us ({gf / us0 * 1e-3:5.1f} TF/s)'
            print(row, flush=True)
            yr = ref64(x, w, rb, 0) if check else None
            dr = ref64(go, w, rb, 1) if check else None
            for H in [int(h) for h in os.environ.get('PROF_RS_H', '320').split(',') if h]:
                os.environ['U3D_RS_H'] = str(H)
                try:
                    with sparse.conv_rs(True):
                        if not sparse._rs_ok(cs, cd, n, rb, P.FMT_X3):
                            continue
                        xg = x.clone().requires_grad_()
                        y1 = sparse.sparse_conv(xg, w, rb)
                        y1.backward(go)
                        d1 = xg.grad
                        us1 = timed(lambda: sparse.sparse_conv(x, w, rb))
                    msg = f'      rs H={H}: {us1:7.1f} us ({gf / us1 * 1e-3:5.1f} TF/s) x{us0 / us1:4.2f}'
                    if check:
                        e = lambda a, r: float((a.detach().double() - r).abs().max() / r.abs().max())
                        msg += f' | fwd err rs {e(y1, yr):.2e} pairs {e(y0, yr):.2e} | dgrad err rs {e(d1, dr):.2e} pairs {e(d0, dr):.2e}'
                    print(msg, flush=True)
                except Exception as ex:      # noqa: BLE001
                    print(f'      rs H={H}: FAILED {ex}', flush=True)
            for T, H in plans:
                os.environ['U3D_TS_T'], os.environ['U3D_TS_H'] = str(T), str(H)
                try:
                    with sparse.conv_ts(True):
                        if sparse._ts_plan(cs, cd, n) is None:
                            continue
                        xg = x.clone().requires_grad_()
                        y1 = sparse.sparse_conv(xg, w, rb)
                        y1.backward(go)
                        d1 = xg.grad
                        us1 = timed(lambda: sparse.sparse_conv(x, w, rb))
                    msg = f'      ts T={T} H={H}: {us1:7.1f} us ({gf / us1 * 1e-3:5.1f} TF/s) x{us0 / us1:4.2f}'
                    if check:
                        e = lambda a, r: float((a.detach().double() - r).abs().max() / r.abs().max())
                        msg += f' | fwd err ts {e(y1, yr):.2e} pairs {e(y0, yr):.2e} | dgrad err ts {e(d1, dr):.2e} pairs {e(d0, dr):.2e}'
                    print(msg, flush=True)
                except Exception as ex:      # noqa: BLE001
                    print(f'      ts T={T} H={H}: FAILED {ex}', flush=True)
                finally:
                    os.environ.pop('U3D_TS_T', None); os.environ.pop('U3D_TS_H', None)

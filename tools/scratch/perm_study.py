import os, sys
import numpy as np, torch
ROOT = '/root/repo'
sys.path.insert(0, ROOT)
from oracle import sparse_ops as O
from unidet3d_amd.synthetic import make_scene
scenes = [make_scene(i) for i in range(2)]
coords, _f, _inv, shape = O.voxelize([torch.from_numpy(s.points) for s in scenes], 0.02, 128)
pairs = O.build_subm_rulebook(coords, shape)
n = coords.shape[0]
valid = np.zeros((27, n), bool)
for k, (i, o) in enumerate(pairs):
    valid[k, o] = True
P = valid.sum()
def cost(order, T):
    # order: permutation within tiles; cost = number of (16-subtile, k) nonempty
    v = valid[:, order]
    n16 = n // 16
    return v[:, :n16 * 16].reshape(27, n16, 16).any(2).sum() * 16 / P
ident = np.arange(n)
print('identity', cost(ident, 128))
maskint = np.zeros(n, np.int64)
for k in range(27):
    maskint |= valid[k].astype(np.int64) << k
for T in (64, 128, 256, 512):
    # (a) sort by mask int within tile
    order = ident.copy()
    for t in range(0, n - T + 1, T):
        seg = order[t:t + T]
        order[t:t + T] = seg[np.argsort(maskint[seg], kind='stable')]
    ca = cost(order, T)
    # (b) recursive median split on most balanced bit
    def split(rows, depth):
        if len(rows) <= 16 or depth == 0:
            return [rows]
        best, bb = None, 1e9
        for k in range(27):
            c = valid[k, rows].sum()
            b = abs(c - len(rows) / 2)
            if b < bb:
                bb, best = b, k
        on = rows[valid[best, rows]]
        off = rows[~valid[best, rows]]
        return split(off, depth - 1) + split(on, depth - 1)
    order2 = ident.copy()
    for t in range(0, n - T + 1, T):
        groups = split(order2[t:t + T].copy(), 6)
        order2[t:t + T] = np.concatenate(groups)
    cb = cost(order2, T)
    # (c) greedy: sort by popcount then mask
    pc = valid.sum(0)
    order3 = ident.copy()
    for t in range(0, n - T + 1, T):
        seg = order3[t:t + T]
        order3[t:t + T] = seg[np.lexsort((maskint[seg], pc[seg]))]
    cc = cost(order3, T)
    print(f'T={T}: sort-by-mask {ca:.2f}  tree-split {cb:.2f}  popcount-sort {cc:.2f}')

"""CPU study for the tile-stationary sparse convolution (round 6): per-tile unique source rows ("halo") of a SubM-3 rulebook in
canonical row order, 16-row sub-tile x offset occupancy, and the (pass, offset) step count for a given LDS capacity.
usage: python tools/halo_stats.py [n_scenes] [levels]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sparse_ops as O  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
maxlv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scenes = [make_scene(i) for i in range(ns)]
coords, _f, _inv, shape = O.voxelize([torch.from_numpy(s.points) for s in scenes], 0.02, 128)


def nbr_table(coords, shape):
    pairs = O.build_subm_rulebook(coords, shape)
    n = coords.shape[0]
    nbr = np.full((27, n), -1, np.int64)
    for k, (i, o) in enumerate(pairs):
        nbr[k, o] = i
    return nbr


for lv in range(1, maxlv + 1):
    nbr = nbr_table(coords, shape)
    n = nbr.shape[1]
    P = int((nbr >= 0).sum())
    print(f'== level {lv}: n={n} pairs/row={P / n:.2f}')
    # 16-row sub-tiles
    n16 = (n + 15) // 16
    pad = np.full((27, n16 * 16), -1, np.int64)
    pad[:, :n] = nbr
    v16 = (pad.reshape(27, n16, 16) >= 0)
    ne = v16.any(2)
    print(f'   16-row tile x offset: nonempty {ne.mean():.3f} ({ne.sum(0).mean():.1f} of 27 per tile), fill inside nonempty {v16.sum() / (ne.sum() * 16):.3f}'
          f' -> dense MFMA work / pair work = {ne.sum() * 16 / P:.2f}')
    for T in (64, 128, 256):
        nt = (n + T - 1) // T
        sizes = np.empty(nt, np.int64)
        for H in (128, 192, 256):
            steps = 0
            rt_items = 0
            for t in range(nt):
                blk = nbr[:, t * T:(t + 1) * T]
                u = np.unique(blk[blk >= 0])
                sizes[t] = len(u)
                loc = np.searchsorted(u, blk)
                loc[blk < 0] = -1
                npass = (len(u) + H - 1) // H
                for p in range(npass):
                    inp = (loc >= p * H) & (loc < (p + 1) * H)
                    steps += int(inp.any(1).sum())
                    w = inp.shape[1]
                    r16 = (w + 15) // 16
                    pp = np.zeros((27, r16 * 16), bool)
                    pp[:, :w] = inp
                    rt_items += int(pp.reshape(27, r16, 16).any(2).sum())
            if H == 128:
                print(f'   T={T}: halo/T mean {sizes.mean() / T:.2f} p50 {np.median(sizes) / T:.2f} p90 {np.quantile(sizes, .9) / T:.2f} max {sizes.max() / T:.2f};'
                      f' sum halo / n = {sizes.sum() / n:.2f}')
            print(f'      H={H}: (pass,k) steps per tile {steps / nt:.1f}; 16-row items / (n16*27*nonempty) = {rt_items / ne.sum():.3f}')
    coords, shape, _pairs = O.build_down_rulebook(coords, shape)

"""CPU ORACLE (test infrastructure, NOT product code) -- sparse-voxel primitives.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``unidet3d_amd/`` imports it.

PARITY PINNING.  The arithmetic restated here lives in third-party packages
that are NOT vendored under /root/reference and are not installable here:
  * spconv-cu120 == 2.3.6 / cumm-cu120 == 0.5.1 (Dockerfile:51,70): SubMConv3d,
    SparseConv3d, SparseInverseConv3d, SparseConvTensor.  Call sites:
    unidet3d/spconv_unet.py:34-72,146-192, unidet3d/unidet3d.py:96-111,353-354.
  * MinkowskiEngine fork daizhirui/MinkowskiEngine@ce930ee (Dockerfile:9):
    batch_sparse_collate + TensorField.sparse() + inverse_mapping.  Call site:
    unidet3d/unidet3d.py:158-174.
  * torch-scatter == 2.1.2 (Dockerfile:31): scatter_mean.  Call sites:
    unidet3d/unidet3d.py:130,333,446.
The reference ships no tests/golden vectors for these, so this part of the
oracle is "parity unpinned" by the reference; it is pinned instead by the
mathematical identities in tests/test_oracle_identities.py (dense conv3d /
conv_transpose3d sampled at active sites, torch.unique, index_add_/bincount).

Canonical order: voxels sorted ascending by
key = ((b*X + x)*Y + y)*Z + z; rulebook pairs grouped by kernel offset
ascending and, within an offset, ascending by output row.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch


# ----------------------------------------------------------------------------
# R1  voxelisation  (unidet3d/unidet3d.py:136-176 -> MinkowskiEngine)
# ----------------------------------------------------------------------------
def point_features(p: torch.Tensor) -> torch.Tensor:
    """feats = hstack(rgb, xyz - mean(xyz))   (unidet3d.py:160)."""
    return torch.hstack((p[:, 3:], p[:, :3] - p[:, :3].mean(0)))


def voxelize(points: List[torch.Tensor], voxel_size: float, min_spatial_shape: int,
             elastic_points: Optional[List[torch.Tensor]] = None):
    """Restates UniDet3D.collate.

    coords = floor((xyz - min) / voxel_size) (ME.utils.batch_sparse_collate floors
    float coordinates before the int cast); batch index prepended; voxel feature =
    unweighted mean of the voxel's point features (TensorField default
    quantization mode); inverse_mapping = point -> voxel row.

    Returns (coords int32 [Nv,4] canonical order, feats f32 [Nv,6],
             inverse int64 [Np], spatial_shape int64 [3]).
    """
    cs, fs = [], []
    for i, p in enumerate(points):
        if elastic_points is None:
            c = (p[:, :3] - p[:, :3].min(0)[0]) / voxel_size          # :159
        else:
            c = elastic_points[i] - elastic_points[i].min(0)[0]       # :164
        ci = torch.floor(c).to(torch.int64)
        cs.append(torch.cat([torch.full((len(p), 1), i, dtype=torch.int64), ci], 1))
        fs.append(point_features(p))
    coords = torch.cat(cs)
    feats = torch.cat(fs).to(torch.float32)
    spatial_shape = torch.clip(coords.max(0)[0][1:] + 1, min_spatial_shape)   # :168-169
    X, Y, Z = [int(v) for v in spatial_shape]
    key = ((coords[:, 0] * X + coords[:, 1]) * Y + coords[:, 2]) * Z + coords[:, 3]
    ukey, inverse = torch.unique(key, sorted=True, return_inverse=True)
    nv = len(ukey)
    cnt = torch.bincount(inverse, minlength=nv).to(torch.float64)
    vf = torch.zeros(nv, feats.shape[1], dtype=torch.float64)
    vf.index_add_(0, inverse, feats.to(torch.float64))
    vf = (vf / cnt[:, None]).to(torch.float32)
    z = ukey % Z
    y = (ukey // Z) % Y
    x = (ukey // (Z * Y)) % X
    b = ukey // (Z * Y * X)
    vcoords = torch.stack([b, x, y, z], 1).to(torch.int32)
    return vcoords, vf, inverse, spatial_shape


# ----------------------------------------------------------------------------
# R2 / R3  rulebooks  (spconv indice-pair generation)
# ----------------------------------------------------------------------------
def _keys(coords: np.ndarray, shape) -> np.ndarray:
    X, Y, Z = [int(s) for s in shape]
    c = coords.astype(np.int64)
    return ((c[:, 0] * X + c[:, 1]) * Y + c[:, 2]) * Z + c[:, 3]


def build_subm_rulebook(coords: torch.Tensor, spatial_shape) -> List[Tuple[np.ndarray, np.ndarray]]:
    """SubMConv3d(k=3, pad=1) pairs (spconv_unet.py:43-56, unidet3d.py:97-103).

    Offset index k = (dx+1)*9 + (dy+1)*3 + (dz+1) (row-major over the three
    spatial dims in indices[:,1:4] order); input = output + (dx,dy,dz)
    (cross-correlation).  Output set = input set.  ``coords`` must already be
    in canonical order.  Returns 27 (in_rows, out_rows) int32 arrays, each
    ascending in out_rows.
    """
    c = coords.numpy().astype(np.int64)
    shape = np.asarray([int(s) for s in spatial_shape], np.int64)
    keys = _keys(c, shape)
    assert np.all(np.diff(keys) > 0), 'coords not in canonical order'
    n = len(c)
    rows = np.arange(n, dtype=np.int64)
    pairs = []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                nb = c.copy()
                nb[:, 1] += dx
                nb[:, 2] += dy
                nb[:, 3] += dz
                ok = np.all((nb[:, 1:] >= 0) & (nb[:, 1:] < shape[None]), 1)
                nk = _keys(nb, shape)
                pos = np.searchsorted(keys, nk)
                pos_c = np.minimum(pos, n - 1)
                hit = ok & (keys[pos_c] == nk)
                pairs.append((pos_c[hit].astype(np.int32), rows[hit].astype(np.int32)))
    return pairs


def build_down_rulebook(coords: torch.Tensor, spatial_shape):
    """SparseConv3d(k=2, s=2) (spconv_unet.py:148-154): out = in >> 1,
    offset k = (x&1)*4 + (y&1)*2 + (z&1); out_shape = floor(in_shape / 2); an input
    whose parent falls outside out_shape (odd extent) is dropped.

    Returns (out_coords int32 [No,4] canonical, out_shape [3],
             8 x (in_rows, out_rows) int32 ascending in both).
    """
    c = coords.numpy().astype(np.int64)
    shape = np.asarray([int(s) for s in spatial_shape], np.int64)
    oshape = shape // 2
    par = c.copy()
    par[:, 1:] >>= 1
    ok = np.all(par[:, 1:] < oshape[None], 1)
    pk = _keys(par, oshape)
    ukeys = np.unique(pk[ok])
    Z, Y, X = int(oshape[2]), int(oshape[1]), int(oshape[0])
    oz = ukeys % Z
    oy = (ukeys // Z) % Y
    ox = (ukeys // (Z * Y)) % X
    ob = ukeys // (Z * Y * X)
    out_coords = torch.from_numpy(np.stack([ob, ox, oy, oz], 1).astype(np.int32))
    orow = np.searchsorted(ukeys, pk)
    off = (c[:, 1] & 1) * 4 + (c[:, 2] & 1) * 2 + (c[:, 3] & 1)
    rows = np.arange(len(c), dtype=np.int64)
    pairs = []
    for k in range(8):
        m = ok & (off == k)
        pairs.append((rows[m].astype(np.int32), orow[m].astype(np.int32)))
    return out_coords, torch.from_numpy(oshape.copy()), pairs


# ----------------------------------------------------------------------------
# K4-K7  gather - GEMM - scatter convolutions (the algorithm spconv's CPU path
# uses: per-offset index_select -> mm -> index_add_)
# ----------------------------------------------------------------------------
def sparse_conv(feats: torch.Tensor, weight: torch.Tensor, pairs, n_out: int,
                inverse: bool = False) -> torch.Tensor:
    """out[o] += W_k . in[i] over rulebook pairs.

    weight layout [C_out, k0, k1, k2, C_in] (spconv 2.x KRSC).  ``inverse``
    swaps the pair roles (SparseInverseConv3d reuses the forward pairs of the
    strided conv, spconv_unet.py:178-183): out[in_row] += W_k . in[out_row].
    """
    cout = weight.shape[0]
    cin = weight.shape[-1]
    w = weight.reshape(cout, -1, cin)
    out = feats.new_zeros(n_out, cout)
    for k, (ir, orow) in enumerate(pairs):
        if len(ir) == 0:
            continue
        src, dst = (orow, ir) if inverse else (ir, orow)
        src = torch.as_tensor(src, dtype=torch.int64)
        dst = torch.as_tensor(dst, dtype=torch.int64)
        out.index_add_(0, dst, feats.index_select(0, src) @ w[:, k, :].t())
    return out


# ----------------------------------------------------------------------------
# K11 / K12  segment mean (torch_scatter.scatter_mean(dim=0) semantics)
# ----------------------------------------------------------------------------
def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: Optional[int] = None) -> torch.Tensor:
    """Empty segment -> 0; count clamped to >= 1 (torch-scatter 2.1.2)."""
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))

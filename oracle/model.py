"""CPU ORACLE (test infrastructure, NOT product code) -- the model restated.

Follows the *structure* of
  unidet3d/spconv_unet.py:13-240   (ResidualBlock, SpConvUNet)
  unidet3d/unidet3d.py:95-134      (input_conv, output_layer, extract_feat)
  unidet3d/encoder.py:8-283        (SelfAttentionLayer, FFN, PredBBox,
                                    UniDet3DEncoder, _bbox_pred_to_bbox)
with the arithmetic of the absent third-party ops taken from sparse_ops.py.
Parameter names equal the reference's ``state_dict`` names so that one set of
weights drives the oracle and the HIP product path.

Pinned by: tests/golden/encoder_golden.npz (generated from the REAL
reference encoder.py by tools/gen_golden_encoder.py) for the decoder;
dense-conv identities for the backbone (parity unpinned by the reference).
"""
from __future__ import annotations

import functools
import itertools
import math
from collections import OrderedDict
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from . import sparse_ops as so


class OSparse:
    """Minimal stand-in for spconv.SparseConvTensor (fields used at
    spconv_unet.py:83-85,88,216-218,229)."""

    def __init__(self, features, indices, spatial_shape, batch_size, books=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = batch_size
        self.books = {} if books is None else books

    def replace_feature(self, f):
        return OSparse(f, self.indices, self.spatial_shape, self.batch_size, self.books)


class OSubMConv3d(nn.Module):
    def __init__(self, cin, cout, kernel_size, indice_key=None):
        super().__init__()
        k = kernel_size
        self.kernel_size = k
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(cout, k, k, k, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x: OSparse) -> OSparse:
        if self.kernel_size == 1:
            w = self.weight.reshape(self.weight.shape[0], -1)
            return x.replace_feature(x.features @ w.t())
        if self.indice_key not in x.books:
            x.books[self.indice_key] = so.build_subm_rulebook(x.indices, x.spatial_shape)
        pairs = x.books[self.indice_key]
        return x.replace_feature(so.sparse_conv(x.features, self.weight, pairs, len(x.features)))


class OSparseConv3d(nn.Module):
    def __init__(self, cin, cout, indice_key):
        super().__init__()
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(cout, 2, 2, 2, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x: OSparse) -> OSparse:
        oc, oshape, pairs = so.build_down_rulebook(x.indices, x.spatial_shape)
        x.books[self.indice_key] = (pairs, x.indices, x.spatial_shape)
        f = so.sparse_conv(x.features, self.weight, pairs, len(oc))
        return OSparse(f, oc, oshape, x.batch_size, x.books)


class OSparseInverseConv3d(nn.Module):
    def __init__(self, cin, cout, indice_key):
        super().__init__()
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(cout, 2, 2, 2, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x: OSparse) -> OSparse:
        pairs, idx, shape = x.books[self.indice_key]
        f = so.sparse_conv(x.features, self.weight, pairs, len(idx), inverse=True)
        return OSparse(f, idx, shape, x.batch_size, x.books)


class OSeq(nn.Sequential):
    """spconv.SparseSequential: dense modules act on .features only."""

    def forward(self, x: OSparse) -> OSparse:
        for m in self:
            if isinstance(m, (OSubMConv3d, OSparseConv3d, OSparseInverseConv3d, OResidualBlock, OSeq)):
                x = m(x)
            else:
                x = x.replace_feature(m(x.features))
        return x


_norm = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)   # spconv_unet.py:119-124


class OResidualBlock(nn.Module):
    """spconv_unet.py:13-91 (normalize_before=True branch :40-56)."""

    def __init__(self, cin, cout, indice_key):
        super().__init__()
        if cin == cout:
            self.i_branch = OSeq(nn.Identity())
        else:
            self.i_branch = OSeq(OSubMConv3d(cin, cout, 1))
        self.conv_branch = OSeq(
            _norm(cin), nn.ReLU(), OSubMConv3d(cin, cout, 3, indice_key),
            _norm(cout), nn.ReLU(), OSubMConv3d(cout, cout, 3, indice_key))

    def forward(self, x: OSparse) -> OSparse:
        identity = OSparse(x.features, x.indices, x.spatial_shape, x.batch_size, x.books)
        out = self.conv_branch(x)
        return out.replace_feature(out.features + self.i_branch(identity).features)


class OSpConvUNet(nn.Module):
    """spconv_unet.py:94-240."""

    def __init__(self, num_planes, block_reps=2, indice_key_id=1):
        super().__init__()
        self.num_planes = num_planes
        c0 = num_planes[0]
        self.blocks = OSeq(OrderedDict(
            (f'block{i}', OResidualBlock(c0, c0, f'subm{indice_key_id}')) for i in range(block_reps)))
        if len(num_planes) > 1:
            c1 = num_planes[1]
            self.conv = OSeq(_norm(c0), nn.ReLU(), OSparseConv3d(c0, c1, f'spconv{indice_key_id}'))
            self.u = OSpConvUNet(num_planes[1:], block_reps, indice_key_id + 1)
            self.deconv = OSeq(_norm(c1), nn.ReLU(), OSparseInverseConv3d(c1, c0, f'spconv{indice_key_id}'))
            self.blocks_tail = OSeq(OrderedDict(
                (f'block{i}', OResidualBlock(c0 * (2 - i), c0, f'subm{indice_key_id}'))
                for i in range(block_reps)))

    def forward(self, x: OSparse, previous_outputs=None):
        out = self.blocks(x)
        identity = out
        if len(self.num_planes) > 1:
            d = self.conv(out)
            d, previous_outputs = self.u(d, previous_outputs)
            d = self.deconv(d)
            out = out.replace_feature(torch.cat((identity.features, d.features), dim=1))
            out = self.blocks_tail(out)
        if previous_outputs is None:
            previous_outputs = []
        previous_outputs.append(out)
        return out, previous_outputs


# ----------------------------------------------------------------------------
# decoder (encoder.py)
# ----------------------------------------------------------------------------
class _OMHA(nn.Module):
    """Parameters of nn.MultiheadAttention(d, h, batch_first=True), math spelled out."""

    def __init__(self, d, h):
        super().__init__()
        self.h = h
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x):
        n, d = x.shape
        hd = d // self.h
        q, k, v = F.linear(x, self.in_proj_weight, self.in_proj_bias).chunk(3, -1)
        q = q.view(n, self.h, hd).transpose(0, 1)
        k = k.view(n, self.h, hd).transpose(0, 1)
        v = v.view(n, self.h, hd).transpose(0, 1)
        a = torch.softmax((q @ k.transpose(1, 2)) / math.sqrt(hd), -1)
        o = (a @ v).transpose(0, 1).reshape(n, d)
        return self.out_proj(o)


class _OAttnLayer(nn.Module):                 # encoder.py:8-41
    def __init__(self, d, h):
        super().__init__()
        self.attn = _OMHA(d, h)
        self.norm = nn.LayerNorm(d)

    def forward(self, xs):
        return [self.norm(self.attn(x) + x) for x in xs]


class _OFFN(nn.Module):                       # encoder.py:43-80
    def __init__(self, d, hidden, act):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(d, hidden), nn.ReLU() if act == 'relu' else nn.GELU(),
                                 nn.Dropout(0.0), nn.Linear(hidden, d), nn.Dropout(0.0))
        self.norm = nn.LayerNorm(d)

    def forward(self, xs):
        return [self.norm(self.net(y) + y) for y in xs]


class _OPredBBox(nn.Module):                  # encoder.py:82-111
    def __init__(self, d, n):
        super().__init__()
        self.linear = nn.Linear(d, n)

    def forward(self, x):
        x = self.linear(x)
        return torch.hstack((torch.exp(x[:, :6]), x[:, 6:]))


def bbox_pred_to_bbox(points, bbox_pred):     # encoder.py:241-283
    if bbox_pred.shape[0] == 0:
        return bbox_pred
    xc = points[:, 0] + (bbox_pred[:, 1] - bbox_pred[:, 0]) / 2
    yc = points[:, 1] + (bbox_pred[:, 3] - bbox_pred[:, 2]) / 2
    zc = points[:, 2] + (bbox_pred[:, 5] - bbox_pred[:, 4]) / 2
    base = torch.stack([xc, yc, zc, bbox_pred[:, 0] + bbox_pred[:, 1],
                        bbox_pred[:, 2] + bbox_pred[:, 3], bbox_pred[:, 4] + bbox_pred[:, 5]], -1)
    if bbox_pred.shape[1] == 6:
        return base
    scale = bbox_pred[:, 0] + bbox_pred[:, 1] + bbox_pred[:, 2] + bbox_pred[:, 3]
    q = torch.exp(torch.sqrt(bbox_pred[:, 6] ** 2 + bbox_pred[:, 7] ** 2))
    alpha = 0.5 * torch.atan2(bbox_pred[:, 6], bbox_pred[:, 7])
    return torch.stack((xc, yc, zc, scale / (1 + q), scale / (1 + q) * q,
                        bbox_pred[:, 5] + bbox_pred[:, 4], alpha), dim=-1)


class OEncoder(nn.Module):                    # encoder.py:113-239
    def __init__(self, num_layers, datasets_classes, in_channels, d_model, num_heads,
                 hidden_dim, dropout, activation_fn, datasets, angles, **kw):
        super().__init__()
        assert dropout == 0.0
        self.num_layers = num_layers
        self.datasets = datasets
        self.angles = angles
        self.input_proj = nn.Sequential(nn.Linear(in_channels, d_model), nn.ReLU(),
                                        nn.Linear(d_model, d_model))
        self.self_attn_layers = nn.ModuleList(_OAttnLayer(d_model, num_heads) for _ in range(num_layers))
        self.ffn_layers = nn.ModuleList(_OFFN(d_model, hidden_dim, activation_fn) for _ in range(num_layers))
        self.out_norm = nn.LayerNorm(d_model)
        unique_cls = sorted(set(itertools.chain.from_iterable(datasets_classes))) + ['no_obj']
        self.outs_cls = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(),
                                      nn.Linear(d_model, len(unique_cls)))
        self.datasets_cls_idxs = [[unique_cls.index(c) for c in dc] + [-1] for dc in datasets_classes]
        self.out_bboxes = _OPredBBox(d_model, 8)

    def _head(self, feats, sp_centers, names):
        cls_preds, boxes = [], []
        for i, f in enumerate(feats):
            nq = self.out_norm(f)
            idx = self.datasets.index(names[i])
            cidx = torch.tensor(self.datasets_cls_idxs[idx], dtype=torch.long)
            cls_preds.append(self.outs_cls(nq)[:, cidx])
            pb = self.out_bboxes(nq)
            if not self.angles[idx]:
                pb = pb[:, :6]
            boxes.append(bbox_pred_to_bbox(sp_centers[i], pb))
        return cls_preds, boxes

    def forward(self, x, sp_centers, names):
        feats = [self.input_proj(y) for y in x]
        outs = [self._head(feats, sp_centers, names)]
        for i in range(self.num_layers):
            feats = self.ffn_layers[i](self.self_attn_layers[i](feats))
            outs.append(self._head(feats, sp_centers, names))
        aux = [dict(cls_preds=c, bboxes=b) for c, b in outs[:-1]]
        return dict(cls_preds=outs[-1][0], bboxes=outs[-1][1], aux_outputs=aux)


# ----------------------------------------------------------------------------
# detector glue (unidet3d.py:95-134, 277-364)
# ----------------------------------------------------------------------------
class ODetector(nn.Module):
    def __init__(self, in_channels=6, num_channels=32, voxel_size=0.02, min_spatial_shape=128,
                 backbone=None, decoder=None):
        super().__init__()
        self.voxel_size = voxel_size
        self.min_spatial_shape = min_spatial_shape
        self.input_conv = OSeq(OSubMConv3d(in_channels, num_channels, 3, 'subm1'))
        self.unet = OSpConvUNet(backbone['num_planes'])
        self.output_layer = OSeq(nn.BatchNorm1d(num_channels, eps=1e-4, momentum=0.1), nn.ReLU())
        self.decoder = OEncoder(**{k: v for k, v in decoder.items() if k != 'type'})

    def extract_feat(self, points: List[torch.Tensor], superpoints: List[torch.Tensor], elastic_coords=None):
        """collate + SparseConvTensor + extract_feat (unidet3d.py:349-357); elastic_coords as in :351."""
        coords, feats, inverse, shape = so.voxelize(points, self.voxel_size, self.min_spatial_shape, elastic_coords)
        feats = feats.to(self.output_layer[0].weight.dtype)      # fp64 run of the oracle (gradient ground truth): same voxels, wider arithmetic
        x = OSparse(feats, coords, shape, len(points))
        x = self.input_conv(x)
        x, _ = self.unet(x)
        x = self.output_layer(x)
        offs, bias, sp = [0], 0, []
        for s in superpoints:
            sp.append(s + bias)
            bias = int(sp[-1].max()) + 1
            offs.append(bias)
        pooled = so.scatter_mean(x.features[inverse], torch.cat(sp), dim_size=bias)
        return [pooled[offs[i]:offs[i + 1]] for i in range(len(points))], x

    def train_points(self, points, elastic_coords=None):          # unidet3d.py:295-302
        """The frame the training targets live in: (elastic - min) * voxel_size when elastic coordinates are given
        (the reference's train pipeline always provides them), else xyz - min."""
        if elastic_coords is not None:
            return [(e - e.min(0)[0]) * self.voxel_size for e in elastic_coords]
        return [p[:, :3] - p[:, :3].min(0)[0] for p in points]

    def sp_centers(self, points, superpoints, elastic_coords=None):   # unidet3d.py:332-333
        return [so.scatter_mean(p, s) for p, s in zip(self.train_points(points, elastic_coords), superpoints)]

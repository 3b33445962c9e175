"""Transformer query decoder over packed variable-length scenes.

Drop-in for the reference's ``UniDet3DEncoder`` (unidet3d/encoder.py:113-239): same registry
name, constructor arguments (:131-133), ``forward(x, sp_centers, datasets_names)`` returning
``dict(cls_preds, bboxes, aux_outputs)`` (:203-239), same ``state_dict`` keys
(``input_proj.{0,2}``, ``self_attn_layers.{i}.attn.{in_proj_weight,in_proj_bias,out_proj.*}``,
``.norm``, ``ffn_layers.{i}.net.{0,3}``, ``.norm``, ``out_norm``, ``outs_cls.{0,2}``,
``out_bboxes.linear``) and public attributes (``datasets``, ``angles``, ``datasets_cls_idxs``).

MI355X-first differences below the surface: the per-scene Python loops of the reference
(:36, :75, :189, :218) are gone -- all scenes are packed into one [sum n_i, d] matrix with
``cu_seqlens``; projections are single library GEMMs over the packed matrix; self-attention is
one varlen flash kernel (include/u3d.h u3d_attn_varlen_*) that never writes the n x n scores.
"""
from __future__ import annotations

import collections.abc
import itertools
import math
import os
from typing import List

import torch
from torch import nn

from . import _lib as L
from . import precision as P
from . import account
from .dense import LayerNorm, linear, ln_linear, mlp
from .ops import cat_views
from .registry import MODELS


_LN_ALIASES = os.environ.get('U3D_LN_ALIASES', '1') != '0'


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_len, H, sum_sq):
        qkv = qkv.contiguous()
        n, d3 = qkv.shape
        d = d3 // 3
        hd = d // H
        b16 = qkv.dtype == torch.bfloat16          # bf16 tensors in and out (precision.bf16_act(): include/u3d.h u3d_attn_varlen_*_b16)
        out = torch.empty(n, d, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(H, n, dtype=torch.float32, device=qkv.device)
        B = cu_seqlens.numel() - 1
        # algorithmic work (bench accounting): S = Q K^T and O = P V -> 4 n_i^2 d flops per scene; qkv read once, out written once
        flops = 4.0 * sum_sq * d if account.ON and sum_sq else 0.0
        if flops:
            account.add('attn_fwd', flops, qkv.element_size() * n * 4 * d + 4.0 * n * H)
        ctx.sum_sq = sum_sq
        ctx.sfx = '_b16' if b16 else ('_bf16' if P.bf16() else '')
        if n:
            L.call('u3d_attn_varlen_fwd' + ctx.sfx, L.ptr(qkv), L.ptr(cu_seqlens), B, max_len, n, H, hd, 1.0 / math.sqrt(hd),
                   L.ptr(out), L.ptr(lse), flops, L.stream())
        ctx.save_for_backward(qkv, out, lse, cu_seqlens)
        ctx.max_len, ctx.H = max_len, H
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        dout = dout.contiguous()
        n, d3 = qkv.shape
        H = ctx.H
        hd = d3 // 3 // H
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(H, n, dtype=torch.float32, device=qkv.device)
        # five matrix products are needed (S, dP, dV, dK, dQ): 10 n_i^2 d flops; qkv, out, dout read, dqkv written
        flops = 10.0 * ctx.sum_sq * (d3 // 3) if account.ON and ctx.sum_sq else 0.0
        if flops:
            account.add('attn_bwd', flops, qkv.element_size() * n * 8 * (d3 // 3) + 4.0 * n * 2 * H)
        if n:
            L.call('u3d_attn_varlen_bwd' + ctx.sfx, L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(cu), cu.numel() - 1,
                   ctx.max_len, n, H, hd, 1.0 / math.sqrt(hd), L.ptr(dqkv), L.ptr(delta), flops, L.stream())
        return dqkv, None, None, None, None


def attention_varlen(qkv, cu_seqlens, max_len, num_heads, sum_sq=0):
    """softmax(Q K^T / sqrt(hd)) V per scene and head; qkv [n, 3*d] packed, cu_seqlens int32 [B+1];
    ``sum_sq`` = sum of n_i^2 (host value, only used for the bench's flops accounting)."""
    return _AttnFn.apply(qkv, cu_seqlens, max_len, num_heads, sum_sq)


class _MHA(nn.Module):
    """Parameters of nn.MultiheadAttention(d, h, batch_first=True) (same names / init)."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x, cu_seqlens, max_len, sum_sq=0):
        qkv = linear(x, self.in_proj_weight, self.in_proj_bias, out_bf16=True)     # a bf16 tensor under precision.bf16_act(), fp32 otherwise
        o = attention_varlen(qkv, cu_seqlens, max_len, self.num_heads, sum_sq)
        return linear(o, self.out_proj.weight, self.out_proj.bias)


class SelfAttentionLayer(nn.Module):          # encoder.py:8-41
    def __init__(self, d_model, num_heads, dropout):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError('the reference configs use dropout=0.0; dropout is not built')
        self.attn = _MHA(d_model, num_heads)
        self.norm = LayerNorm(d_model)

    def forward(self, x, cu_seqlens, max_len, sum_sq=0, res=None, n_out=1):
        """LayerNorm(attn(x) + res), the add inside the LayerNorm kernel.  ``res``: the residual (default x) -- the encoder hands the
        attention and the residual their own alias of the previous LayerNorm's result; ``n_out``: how many aliases of THIS result to
        return (one per consumer: their gradients are summed in the LayerNorm's backward kernel, dense.layer_norm)."""
        return self.norm(self.attn(x, cu_seqlens, max_len, sum_sq), x if res is None else res, n_out)


class FFN(nn.Module):                         # encoder.py:43-80
    def __init__(self, d_model, hidden_dim, dropout, activation_fn):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(d_model, hidden_dim), nn.ReLU() if activation_fn == 'relu' else nn.GELU(),
                                 nn.Dropout(dropout), nn.Linear(hidden_dim, d_model), nn.Dropout(dropout))
        self.norm = LayerNorm(d_model)
        self.act = 'relu' if activation_fn == 'relu' else 'gelu'

    def forward(self, x, res=None, n_out=1):
        # bias + activation in the first GEMM's epilogue, residual add inside the LayerNorm kernel (``res`` / ``n_out``: SelfAttentionLayer)
        return self.norm(mlp(x, self.net[0].weight, self.net[0].bias, self.net[3].weight, self.net[3].bias, self.act), x if res is None else res, n_out)


class PredBBox(nn.Module):                    # encoder.py:82-111
    def __init__(self, d_model, n_bbox_outs, bbox_init_normal=False):
        super().__init__()
        self.linear = nn.Linear(d_model, n_bbox_outs)
        if bbox_init_normal:
            nn.init.normal_(self.linear.weight, std=.01)

    @staticmethod
    def decode(x):                                # encoder.py:108-111
        return torch.hstack((torch.exp(x[:, :6]), x[:, 6:]))

    def forward(self, x):
        return self.decode(linear(x, self.linear.weight, self.linear.bias))


class _LazyColumns(collections.abc.Sequence):
    """A read-only list of tensors whose entries are computed on first access: ``thunks[i]()`` makes entry i, once.  Slices are views
    onto the same thunks and the same cache, so ``cols[a:b]`` costs nothing until somebody reads an entry."""

    def __init__(self, thunks, _cache=None, _idx=None):
        self._thunks = thunks
        self._cache = [None] * len(thunks) if _cache is None else _cache
        self._idx = list(range(len(thunks))) if _idx is None else _idx

    def __len__(self):
        return len(self._idx)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return _LazyColumns(self._thunks, self._cache, self._idx[i])
        j = self._idx[i]
        if self._cache[j] is None:
            self._cache[j] = self._thunks[j]()
        return self._cache[j]


class _BoxDecodeFn(torch.autograd.Function):
    """PredBBox's exp + _bbox_pred_to_bbox for a yaw-free head as one kernel each way (include/u3d.h u3d_box_decode_*)."""

    @staticmethod
    def forward(ctx, raw, centers):
        raw, centers = raw.contiguous(), centers.contiguous()
        box = torch.empty(raw.shape[0], 6, dtype=torch.float32, device=raw.device)
        L.call('u3d_box_decode_fwd', L.ptr(raw), L.ptr(centers), raw.shape[0], L.ptr(box), L.stream())
        ctx.save_for_backward(raw)
        return box

    @staticmethod
    def backward(ctx, dbox):
        raw, = ctx.saved_tensors
        draw = torch.empty_like(raw)
        L.call('u3d_box_decode_bwd', L.ptr(raw), L.ptr(dbox.contiguous()), raw.shape[0], L.ptr(draw), L.stream())
        return draw, None


class _BoxDecode7Fn(torch.autograd.Function):
    """PredBBox's exp + _bbox_pred_to_bbox of a head with a heading -- or of a mixed batch: ``yaw_rows`` (uint8 [M] or None = all) marks
    the rows that have one, the others come out as (centre, size, 0) -- as one kernel each way (u3d_box_decode7_*)."""

    @staticmethod
    def forward(ctx, raw, centers, yaw_rows):
        raw, centers = raw.contiguous(), centers.contiguous()
        box = torch.empty(raw.shape[0], 7, dtype=torch.float32, device=raw.device)
        L.call('u3d_box_decode7_fwd', L.ptr(raw), L.ptr(centers), L.ptr(yaw_rows), raw.shape[0], L.ptr(box), L.stream())
        ctx.save_for_backward(raw)
        ctx.yaw_rows = yaw_rows
        return box

    @staticmethod
    def backward(ctx, dbox):
        raw, = ctx.saved_tensors
        draw = torch.empty_like(raw)
        L.call('u3d_box_decode7_bwd', L.ptr(raw), L.ptr(dbox.contiguous()), L.ptr(ctx.yaw_rows), raw.shape[0], L.ptr(draw), L.stream())
        return draw, None, None


def _bbox_pred_to_bbox(points, bbox_pred):    # encoder.py:241-283
    if bbox_pred.shape[0] == 0:
        return bbox_pred
    half = (bbox_pred[:, 1:6:2] - bbox_pred[:, 0:6:2]) / 2            # (max - min) / 2 per axis
    center = points + half
    size = bbox_pred[:, 0:6:2] + bbox_pred[:, 1:6:2]
    if bbox_pred.shape[1] == 6:
        return torch.cat((center, size), dim=-1)
    scale = bbox_pred[:, 0] + bbox_pred[:, 1] + bbox_pred[:, 2] + bbox_pred[:, 3]
    q = torch.exp(torch.sqrt(bbox_pred[:, 6] ** 2 + bbox_pred[:, 7] ** 2))
    alpha = 0.5 * torch.atan2(bbox_pred[:, 6], bbox_pred[:, 7])
    return torch.cat((center, (scale / (1 + q))[:, None], (scale / (1 + q) * q)[:, None], size[:, 2:3],
                      alpha[:, None]), dim=-1)


@MODELS.register_module()
class UniDet3DEncoder(nn.Module):
    def __init__(self, num_layers, datasets_classes, in_channels, d_model, num_heads, hidden_dim, dropout,
                 activation_fn, datasets, angles, **kwargs):
        super().__init__()
        self.num_layers = num_layers
        self.datasets = datasets
        self.angles = angles
        self.num_heads = num_heads
        self.input_proj = nn.Sequential(nn.Linear(in_channels, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.self_attn_layers = nn.ModuleList(SelfAttentionLayer(d_model, num_heads, dropout) for _ in range(num_layers))
        self.ffn_layers = nn.ModuleList(FFN(d_model, hidden_dim, dropout, activation_fn) for _ in range(num_layers))
        self.out_norm = LayerNorm(d_model)
        unique_cls = sorted(set(itertools.chain.from_iterable(datasets_classes))) + ['no_obj']
        self.outs_cls = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, len(unique_cls)))
        self.datasets_cls_idxs = [[unique_cls.index(c) for c in dc] + [-1] for dc in datasets_classes]
        self.out_bboxes = PredBBox(d_model, 8)
        self._cidx_cache = {}

    def _cidx(self, idx, device):
        key = (idx, str(device))
        if key not in self._cidx_cache:      # class-column indices live on the device once: no per-call H2D copy
            self._cidx_cache[key] = torch.as_tensor(self.datasets_cls_idxs[idx], dtype=torch.long, device=device)
        return self._cidx_cache[key]

    def _forward_head(self, feats, sizes, sp_centers, centers_packed, datasets_names):
        """Packed head: one LayerNorm / class MLP / box Linear over all scenes (encoder.py:165-201).
        With a single dataset in the batch the class-column select and the box decode also run once
        on the packed matrix and the per-scene outputs are views of it."""
        # out_norm -> out_bboxes.linear as one op (u3d_ln_linear); nq also feeds the class MLP
        nq, box_raw = ln_linear(feats, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, self.out_bboxes.linear.weight,
                                self.out_bboxes.linear.bias)
        w1, b1, w2, b2 = self.outs_cls[0].weight, self.outs_cls[0].bias, self.outs_cls[2].weight, self.outs_cls[2].bias
        single = len(set(datasets_names)) == 1
        yaw_free = single and not self.angles[self.datasets.index(datasets_names[0])]
        if single:
            # the dataset's class columns (encoder.py:192-194) are selected as ROWS of the [n_cls, d] output weight: the packed
            # [sum n_i, n_cls] logit matrix is never gathered (nor scattered back in backward)
            idx = self.datasets.index(datasets_names[0])
            cidx = self._cidx(idx, feats.device)
            cls_p = mlp(nq, w1, b1, w2[cidx], b2[cidx], 'relu')
            box_p = _BoxDecodeFn.apply(box_raw, centers_packed) if yaw_free else _BoxDecode7Fn.apply(box_raw, centers_packed, None)
            return list(cls_p.split(sizes)), list(box_p.split(sizes)), (cls_p, box_p)
        # mixed batch (joint config): full-width logits once, and ONE decode of the packed box matrix (u3d_box_decode7_*): rows of scenes
        # whose dataset has a heading come out 7-dof, the others as (centre, size, 0) -- the reference never evaluates the heading
        # columns of a yaw-free dataset's scenes (encoder.py:186-199), and neither does the kernel (zero gradient there) -- the
        # per-scene outputs of the reference's dict contract are row / column slices of it (the reference decodes scene by scene)
        cls_all = mlp(nq, w1, b1, w2, b2, 'relu')
        idxs = [self.datasets.index(name) for name in datasets_names]
        any_yaw = any(self.angles[i] for i in idxs)
        if any_yaw:
            flags = L.h2d([int(bool(self.angles[i])) for i in idxs], torch.uint8, feats.device)
            yaw_rows = torch.repeat_interleave(flags, L.h2d(list(sizes), torch.int64, feats.device), output_size=int(sum(sizes)))
            box_p = _BoxDecode7Fn.apply(box_raw, centers_packed, yaw_rows)          # [M, 7] for the criterion kernel
        else:
            box_p = _BoxDecodeFn.apply(box_raw, centers_packed)                     # [M, 6]
        # The per-scene class-column selection (encoder.py:192-194) is an index launch per scene and head application -- 56 per step of
        # the joint config -- whose results the batched criterion never reads (it takes cls_all and the column lists): the list
        # computes an entry when somebody asks for it (predict, the per-scene criterion path, tests).
        cls_rows = cls_all.split(sizes)
        cls_preds = _LazyColumns([(lambda c=c, idx=idx: c[:, self._cidx(idx, feats.device)]) for c, idx in zip(cls_rows, idxs)])
        boxes = []
        for pb, raw, idx in zip(box_p.split(sizes), box_raw.split(sizes), idxs):
            if raw.shape[0] == 0:       # the reference returns an EMPTY scene's 8 (or 6) raw columns undecoded (encoder.py:253-254)
                boxes.append(raw if self.angles[idx] else raw[:, :6])
            else:
                boxes.append(pb if self.angles[idx] or not any_yaw else pb[:, :6])
        return cls_preds, boxes, (cls_all, box_p)

    def forward(self, x: List[torch.Tensor], sp_centers: List[torch.Tensor], datasets_names: List[str]):
        sizes = [int(t.shape[0]) for t in x]
        dev = x[0].device
        cu = L.h2d([0] + list(itertools.accumulate(sizes)), torch.int32, dev)
        max_len = max(sizes) if sizes else 0
        sum_sq = sum(s * s for s in sizes)
        # the per-scene lists are row slices of one packed tensor when they come from UniDet3D.extract_feat: use it as it is (no cat
        # kernel forward, no 8 x (zero-fill + copy) + 7 adds backward)
        centers_packed = cat_views(sp_centers)
        x0 = cat_views(x)
        feats = mlp(x0, self.input_proj[0].weight, self.input_proj[0].bias, self.input_proj[2].weight, self.input_proj[2].bias, 'relu')
        layer_feats = [feats]
        # A LayerNorm result with several consumers (the next Linear, the residual into the next LayerNorm, the head below) is handed out
        # as one alias per consumer: the gradients then reach the LayerNorm's backward kernel separately and are summed there as they are
        # read -- 16 of autograd's 18 elementwise gradient sums per step (cfg2: 52 MB each) are gone.
        x_in = r_in = feats
        for i in range(self.num_layers):
            if not _LN_ALIASES:                  # U3D_LN_ALIASES=0: one tensor per result, autograd sums (A/B runs)
                feats = self.ffn_layers[i](self.self_attn_layers[i](feats, cu, max_len, sum_sq))
                layer_feats.append(feats)
                continue
            a_in, a_res = self.self_attn_layers[i](x_in, cu, max_len, sum_sq, res=r_in, n_out=2)
            if i + 1 < self.num_layers:
                x_in, r_in, feats = self.ffn_layers[i](a_in, res=a_res, n_out=3)
            else:
                feats = self.ffn_layers[i](a_in, res=a_res)
            layer_feats.append(feats)
        # The reference applies the shared prediction head after the input projection and after every layer (encoder.py:221-239).
        # The head is row-wise (LayerNorm, Linear, ReLU), so its num_layers + 1 applications on [n, d] are ONE application on the
        # [(num_layers + 1) n, d] concatenation: a seventh of the launches, GEMMs seven times taller (fewer half-empty waves of
        # workgroups) and one weight-gradient GEMM per head weight instead of seven that autograd has to sum.  Final layer
        # first -- the order in which the criterion stacks the layers.
        B, NL = len(sizes), len(layer_feats)
        order = [layer_feats[-1]] + layer_feats[:-1]
        cls_l, box_l, packed = self._forward_head(torch.cat(order), sizes * NL, list(sp_centers) * NL,
                                                  centers_packed.repeat(NL, 1), list(datasets_names) * NL)
        outs = [(cls_l[j * B:(j + 1) * B], box_l[j * B:(j + 1) * B]) for j in range(NL)]
        res = dict(cls_preds=outs[0][0], bboxes=outs[0][1], aux_outputs=[dict(cls_preds=c, bboxes=b) for c, b in outs[1:]])
        if packed is not None:
            # packed views of the same tensors ([L, sum n_i, .], final layer first), consumed by the criterion's batched path;
            # ignored by anything that follows the reference's dict contract
            n = sum(sizes)
            cls_p, box_p = packed
            res['_packed'] = dict(cls=[cls_p[j * n:(j + 1) * n] for j in range(NL)], box=[box_p[j * n:(j + 1) * n] for j in range(NL)],
                                  sizes=sizes, cls_stacked=cls_p.view(NL, n, cls_p.shape[-1]), box_stacked=box_p.view(NL, n, box_p.shape[-1]))
            if len(set(datasets_names)) > 1:        # mixed batch: which columns / box form belong to which scene
                CU = cls_p.shape[-1]
                idxs = [self.datasets.index(name) for name in datasets_names]
                res['_packed'].update(cidx=[[c if c >= 0 else CU + c for c in self.datasets_cls_idxs[i]] for i in idxs],
                                      yaw=[bool(self.angles[i]) for i in idxs])
        return res

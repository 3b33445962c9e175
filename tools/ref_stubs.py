"""Import-time stand-ins that let the REAL reference files be imported in the build container.

Runs ONLY here (needs /root/reference); nothing under ``unidet3d_amd/`` or the GPU tests imports this.
The reference's files import mmengine / mmdet / mmdet3d / mmcv / spconv / torch_scatter / terminaltables,
none of which is installed.  Two kinds of stand-in, and every golden file records which of its results
passed through the second kind:

  (A) arithmetic-free: registries, ``BaseModule``, ``BaseTransform``, ``InstanceData``, ``print_log``,
      ``AsciiTable``, module containers of spconv (parameter shapes only) -- containers and plumbing.
  (B) restated third-party arithmetic (listed in ``STUBBED_ARITHMETIC``): mmdet3d's axis-aligned 3-D IoU,
      mmdet's ``weighted_loss`` reduction wrapper, mmcv's rotated-rectangle intersection (taken from
      ``oracle.rotated_iou``), torch_scatter's ``scatter_mean``, mmdet3d's ``rotation_3d_in_axis`` and the
      3-D box IoU used by evaluation.  A golden produced through (B) pins the REFERENCE's code around the
      call (matching, target assignment, reductions, dataset weights, AP protocol ...) but not (B) itself.
"""
from __future__ import annotations

import functools
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF_ROOT = '/root/reference'
REF = os.path.join(REF_ROOT, 'unidet3d')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STUBBED_ARITHMETIC = {
    'mmdet3d.AxisAlignedBboxOverlaps3D': 'mmdet3d 1.4.0 axis_aligned_bbox_overlaps_3d(is_aligned=True, eps=1e-6), restated',
    'mmdet.weighted_loss': 'mmdet 3.3.0 weighted_loss / weight_reduce_loss, restated',
    'mmcv.diff_iou_rotated': 'mmcv@780ffed box2corners / oriented_box_intersection_2d, restated in oracle/rotated_iou.py',
    'torch_scatter.scatter_mean': 'torch-scatter 2.1.2 scatter_mean == sum / clamp(count, 1), restated',
    'mmdet3d.rotation_3d_in_axis': 'mmdet3d 1.4.0 rotation_3d_in_axis(axis=2), restated',
    'mmdet3d.boxes.overlaps': 'mmdet3d BaseInstance3DBoxes.overlaps (3-D IoU), restated for yaw = 0 boxes',
    'spconv.weight_shape': 'spconv 2.3.6 conv weight shape [C_out, k, k, k, C_in] (parameter container only)',
}


class Registry:
    """mmengine.Registry subset: register_module() decorator + build(cfg dict)."""

    def __init__(self, name):
        self.name, self.modules = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.modules[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.modules[cfg.pop('type')](**cfg)


class BaseDataElement:
    def __init__(self, *, metainfo=None, **kwargs):
        object.__setattr__(self, '_metainfo_fields', set())
        object.__setattr__(self, '_data_fields', set())
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if name not in ('_metainfo_fields', '_data_fields'):
            self._data_fields.add(name)
        object.__setattr__(self, name, value)

    def values(self):
        return [getattr(self, k) for k in sorted(self._data_fields)]


class InstanceData(BaseDataElement):
    def __len__(self):
        for k in self._data_fields:
            return len(getattr(self, k))
        return 0


class Boxes:
    """Container for what the reference reads from DepthInstance3DBoxes: tensor (bottom-centre form), gravity_center,
    with_yaw, len, indexing, new_box, convert_to, overlaps.  Built from gravity centres (origin (.5,.5,.5))."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        t = torch.as_tensor(tensor, dtype=torch.float32).reshape(-1, box_dim).clone()
        if not with_yaw and box_dim == 6:
            pass
        if origin != (0.5, 0.5, 0):
            t[:, :3] += t[:, 3:6] * (t.new_tensor((0.5, 0.5, 0)) - t.new_tensor(origin))
        self.tensor, self.box_dim, self.with_yaw = t, box_dim, with_yaw

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = c[:, 2] + self.tensor[:, 5] * 0.5
        return c

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, idx):
        b = Boxes.__new__(Boxes)
        t = self.tensor[idx]
        b.tensor = t.reshape(1, -1) if t.dim() == 1 else t
        b.box_dim, b.with_yaw = self.box_dim, self.with_yaw
        return b

    def new_box(self, data):
        b = Boxes.__new__(Boxes)
        b.tensor = torch.as_tensor(data, dtype=torch.float32)
        b.box_dim, b.with_yaw = self.box_dim, self.with_yaw
        return b

    def convert_to(self, mode):
        return self

    @classmethod
    def overlaps(cls, b1, b2):
        """3-D IoU [n1, n2] of yaw-free boxes in bottom-centre form (x, y, z_bottom, dx, dy, dz[, 0])."""
        a, b = b1.tensor.double(), b2.tensor.double()
        lo1 = torch.stack((a[:, 0] - a[:, 3] / 2, a[:, 1] - a[:, 4] / 2, a[:, 2]), 1)
        hi1 = torch.stack((a[:, 0] + a[:, 3] / 2, a[:, 1] + a[:, 4] / 2, a[:, 2] + a[:, 5]), 1)
        lo2 = torch.stack((b[:, 0] - b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 2]), 1)
        hi2 = torch.stack((b[:, 0] + b[:, 3] / 2, b[:, 1] + b[:, 4] / 2, b[:, 2] + b[:, 5]), 1)
        wh = (torch.min(hi1[:, None], hi2[None]) - torch.max(lo1[:, None], lo2[None])).clamp(min=0)
        inter = wh.prod(-1)
        v1, v2 = (hi1 - lo1).prod(-1), (hi2 - lo2).prod(-1)
        return (inter / (v1[:, None] + v2[None] - inter).clamp(min=1e-8)).float()


def axis_aligned_bbox_overlaps_3d(b1, b2, mode='iou', is_aligned=False, eps=1e-6):
    assert is_aligned and mode == 'iou'
    a1 = (b1[..., 3] - b1[..., 0]) * (b1[..., 4] - b1[..., 1]) * (b1[..., 5] - b1[..., 2])
    a2 = (b2[..., 3] - b2[..., 0]) * (b2[..., 4] - b2[..., 1]) * (b2[..., 5] - b2[..., 2])
    lt = torch.max(b1[..., :3], b2[..., :3])
    rb = torch.min(b1[..., 3:], b2[..., 3:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
    union = torch.max(a1 + a2 - overlap, overlap.new_tensor([eps]))
    return overlap / union


class AxisAlignedBboxOverlaps3D:
    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        return axis_aligned_bbox_overlaps_3d(bboxes1, bboxes2, mode, is_aligned)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    if reduction == 'mean':
        return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def weighted_loss(loss_func):
    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        return weight_reduce_loss(loss_func(pred, target, **kwargs), weight, reduction, avg_factor)
    return wrapper


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    index = torch.as_tensor(index).long()
    dim = dim % src.dim()
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    ix = index.reshape([-1 if d == dim else 1 for d in range(src.dim())]).expand_as(src)
    s = torch.zeros(shape, dtype=src.dtype).scatter_add_(dim, ix, src)
    cnt = torch.zeros(n, dtype=src.dtype).scatter_add_(0, index, torch.ones(len(index), dtype=src.dtype)).clamp(min=1)
    return s / cnt.reshape([-1 if d == dim else 1 for d in range(src.dim())])


def rotation_3d_in_axis(points, angles, axis=0, return_mat=False, clockwise=False):
    assert axis in (2, -1) and points.shape[-1] == 3 and not return_mat and not clockwise
    s, c = torch.sin(angles), torch.cos(angles)
    one, zero = torch.ones_like(c), torch.zeros_like(c)
    rot_T = torch.stack([torch.stack([c, s, zero]), torch.stack([-s, c, zero]), torch.stack([zero, zero, one])])
    return torch.einsum('aij,jka->aik', points, rot_T)


# ---- spconv parameter containers (no arithmetic: shapes + names only) ---------------------------------------
class _SparseModule(nn.Module):
    pass


class _SparseSequential(_SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)


class _Conv(_SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kw):
        super().__init__()
        k = kernel_size
        self.weight = nn.Parameter(torch.zeros(out_channels, k, k, k, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))


class _InvConv(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, **kw):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias)


class _BaseTransform:
    def __call__(self, results):
        return self.transform(results)


class _PointSample(_BaseTransform):
    def __init__(self, num_points, sample_range=None, replace=False):
        self.num_points, self.sample_range, self.replace = num_points, sample_range, replace


class _AsciiTable:
    def __init__(self, data):
        self.table_data = data
        self.inner_footing_row_border = False

    @property
    def table(self):
        return '\n'.join(' | '.join(map(str, r)) for r in self.table_data)


MODELS, TASK_UTILS, TRANSFORMS, METRICS = Registry('model'), Registry('task util'), Registry('transform'), Registry('metric')


def install():
    """Put the stand-ins into sys.modules and register the reference package root as ``ref_unidet3d`` WITHOUT running its
    ``__init__`` (which imports spconv / MinkowskiEngine)."""
    from oracle import rotated_iou as orot

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('mmengine')
    mod('mmengine.model', BaseModule=nn.Module)
    mod('mmengine.structures', InstanceData=InstanceData, BaseDataElement=BaseDataElement)
    mod('mmengine.logging', print_log=lambda *a, **k: None, MMLogger=object)
    mod('mmengine.evaluator', BaseMetric=object)
    mod('mmdet3d')
    mod('mmdet3d.registry', MODELS=MODELS, TASK_UTILS=TASK_UTILS, TRANSFORMS=TRANSFORMS, METRICS=METRICS)
    mod('mmdet3d.models',
        axis_aligned_iou_loss=weighted_loss(lambda pred, target: 1 - axis_aligned_bbox_overlaps_3d(pred, target, is_aligned=True)),
        rotated_iou_3d_loss=weighted_loss(lambda pred, target: 1 - _rot_iou(pred, target)))
    mod('mmdet3d.structures', AxisAlignedBboxOverlaps3D=AxisAlignedBboxOverlaps3D, DepthInstance3DBoxes=Boxes,
        rotation_3d_in_axis=rotation_3d_in_axis)
    mod('mmdet3d.structures.bbox_3d', rotation_3d_in_axis=rotation_3d_in_axis)
    mod('mmdet3d.datasets')
    mod('mmdet3d.datasets.transforms', PointSample=_PointSample)
    mod('mmdet')
    mod('mmdet.models')
    mod('mmdet.models.losses')
    mod('mmdet.models.losses.utils', weighted_loss=weighted_loss)
    mod('mmcv')
    mod('mmcv.ops')
    mod('mmcv.ops.diff_iou_rotated', box2corners=orot.box2corners,
        oriented_box_intersection_2d=lambda c1, c2: (orot.oriented_box_intersection_2d(c1, c2), None))
    mod('mmcv.transforms', BaseTransform=_BaseTransform)
    mod('torch_scatter', scatter_mean=scatter_mean)
    mod('terminaltables', AsciiTable=_AsciiTable)
    sp = mod('spconv')
    spt = mod('spconv.pytorch', SparseSequential=_SparseSequential, SubMConv3d=_Conv, SparseConv3d=_Conv,
              SparseInverseConv3d=_InvConv, SparseModule=_SparseModule, SparseConvTensor=object)
    sp.pytorch = spt
    mod('spconv.pytorch.modules', SparseModule=_SparseModule)
    pkg = types.ModuleType('ref_unidet3d')
    pkg.__path__ = [REF]
    sys.modules['ref_unidet3d'] = pkg


def _rot_iou(pred, target):
    from unidet3d_amd.criterion import diff_iou_rotated_3d       # only reached by mode='iou', unused by the configs
    return diff_iou_rotated_3d(pred, target, False)


def ref(name):
    """Import a reference file as ``ref_unidet3d.<name>`` (relative imports inside it resolve to the other reference files)."""
    return importlib.import_module('ref_unidet3d.' + name)


def extract_defs(path, names, ns=None, cls=None):
    """AST-extract top-level functions (or methods of class ``cls``) from a reference file that cannot be imported whole."""
    import ast
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    ns = {'torch': torch, 'np': np} if ns is None else ns
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns

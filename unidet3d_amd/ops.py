"""Voxelisation and superpoint pooling through the C ABI (host side).

``voxelize``     replaces ME.utils.batch_sparse_collate + ME.TensorField(...).sparse()
                 + inverse_mapping                       (unidet3d/unidet3d.py:158-174)
``PoolPlan`` / ``superpoint_pool``   replace  scatter_mean(x.features[inverse_mapping], superpoints)
                                                          (unidet3d/unidet3d.py:130)
``superpoint_centers``               replaces scatter_mean(points, sp_pts_mask)   (:332-333, :446-447)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib as L
from .sparse import OccupancyIndex


@dataclass
class VoxelBatch:
    coords: torch.Tensor          # int32 [Nv, 4]  (b, x, y, z), canonical order
    feats: torch.Tensor           # f32   [Nv, 6]
    inverse: torch.Tensor         # int64 [Np]     point -> voxel row
    spatial_shape: List[int]
    index: OccupancyIndex
    vox_offsets: torch.Tensor     # int32 [Nv+1]   CSR of points per voxel
    vox_points: torch.Tensor      # int32 [Np]
    pt_offsets: torch.Tensor      # int64 [B+1] (device)
    stats: torch.Tensor           # f32 [B, 12] min, max, mean_xyz, pad
    points: torch.Tensor          # f32 [Np, 6] concatenated
    coord_src: Optional[torch.Tensor] = None      # f32 [Np, 3] elastic coordinates (voxel units) when the batch has them


def voxelize(points: List[torch.Tensor], voxel_size: float, min_spatial_shape: int,
             elastic_points: Optional[List[torch.Tensor]] = None, div_mode: int = 0) -> VoxelBatch:
    dev = points[0].device
    B = len(points)
    sizes = [int(p.shape[0]) for p in points]
    pts = torch.cat([p.to(torch.float32) for p in points]).contiguous() if B > 1 else points[0].to(torch.float32).contiguous()
    n_pts = pts.shape[0]
    offs_h = [0]
    for s in sizes:
        offs_h.append(offs_h[-1] + s)
    offs = torch.tensor(offs_h, dtype=torch.int64, device=dev)
    csrc = None
    vs = float(voxel_size)
    if elastic_points is not None:
        csrc = torch.cat([e.to(torch.float32) for e in elastic_points]).contiguous()
        vs = 1.0
    stats = torch.empty(B, 12, dtype=torch.float32, device=dev)
    gmax = torch.empty(3, dtype=torch.int32, device=dev)
    w = L.ws(L.lib().u3d_vox_scene_stats_ws_bytes(B), dev)
    L.call('u3d_vox_scene_stats', L.ptr(pts), L.ptr(csrc), L.ptr(offs), B, max(sizes), vs, div_mode,
           L.ptr(stats), L.ptr(gmax), L.ptr(w), L.stream())
    shape = [max(int(v) + 1, int(min_spatial_shape)) for v in gmax.tolist()]       # read-back #1
    pt_cell = torch.empty(n_pts, dtype=torch.int64, device=dev)
    if OccupancyIndex.wants_hash(B, shape):
        # large extent: cell ids only, then sort -> unique -> hash table (memory follows occupancy, csrc/hashidx.hip)
        L.call('u3d_vox_mark', L.ptr(pts), L.ptr(csrc), L.ptr(offs), B, max(sizes), L.ptr(stats), vs, div_mode,
               *shape, None, L.ptr(pt_cell), L.stream())
        index = OccupancyIndex.from_cells(pt_cell, B, shape)
    else:
        index = OccupancyIndex.alloc(B, shape, dev)
        L.call('u3d_vox_mark', L.ptr(pts), L.ptr(csrc), L.ptr(offs), B, max(sizes), L.ptr(stats), vs, div_mode,
               *shape, L.ptr(index.bitmap), L.ptr(pt_cell), L.stream())
        index.build_rank()
    n_vox = index.count()                                                          # read-back #2
    coords = index.coords(n_vox)
    inverse = torch.empty(n_pts, dtype=torch.int64, device=dev)
    vox_offsets = torch.empty(n_vox + 1, dtype=torch.int32, device=dev)
    vox_points = torch.empty(n_pts, dtype=torch.int32, device=dev)
    feats = torch.empty(n_vox, 6, dtype=torch.float32, device=dev)
    w2 = L.ws(L.lib().u3d_vox_finalize_ws_bytes(n_pts, n_vox), dev)
    L.call('u3d_vox_finalize', L.ptr(pts), L.ptr(offs), B, n_pts, L.ptr(stats), L.ptr(pt_cell), *index.table(), n_vox, L.ptr(inverse), L.ptr(vox_offsets), L.ptr(vox_points), L.ptr(feats), 6,
           L.ptr(w2), L.stream())
    return VoxelBatch(coords, feats, inverse, shape, index, vox_offsets, vox_points, offs, stats, pts, csrc)


def csr_build(seg_ids: torch.Tensor, S: int):
    """offsets int32 [S+1], list int32 [L] of element ids grouped by segment."""
    Ln = seg_ids.shape[0]
    dev = seg_ids.device
    offsets = torch.empty(S + 1, dtype=torch.int32, device=dev)
    lst = torch.empty(Ln, dtype=torch.int32, device=dev)
    w = L.ws(L.lib().u3d_csr_build_ws_bytes(Ln, S), dev)
    L.call('u3d_csr_build', L.ptr(seg_ids.contiguous()), Ln, S, L.ptr(offsets), L.ptr(lst), L.ptr(w), L.stream())
    return offsets, lst


class PoolPlan:
    """Index structures of one batch for pooling voxel features into superpoints (built once,
    integer work only): CSR of points per superpoint composed with inverse_mapping (forward) and
    the voxelizer's CSR of points per voxel composed with the superpoint ids (backward)."""

    def __init__(self, vb: VoxelBatch, superpoints: torch.Tensor, n_superpoints: int):
        self.S = int(n_superpoints)
        self.n_vox = vb.coords.shape[0]
        sp = superpoints.contiguous()
        self.sp_offsets, self.sp_points = csr_build(sp, self.S)
        n = sp.shape[0]
        self.sp_vox = torch.empty(n, dtype=torch.int32, device=sp.device)
        L.call('u3d_gather_i64_to_i32', L.ptr(vb.inverse), L.ptr(self.sp_points), n, L.ptr(self.sp_vox), L.stream())
        self.vox_sp = torch.empty(n, dtype=torch.int32, device=sp.device)
        L.call('u3d_gather_i64_to_i32', L.ptr(sp), L.ptr(vb.vox_points), n, L.ptr(self.vox_sp), L.stream())
        self.vox_offsets = vb.vox_offsets


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, plan: PoolPlan):
        feats = feats.contiguous()
        C = feats.shape[1]
        out = torch.empty(plan.S, C, dtype=torch.float32, device=feats.device)
        L.call('u3d_segment_gather_sum', L.ptr(feats), L.ptr(plan.sp_vox), L.ptr(plan.sp_offsets), plan.S, C, 1,
               None, L.ptr(out), L.stream())
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        dout = dout.contiguous()
        C = dout.shape[1]
        df = torch.empty(plan.n_vox, C, dtype=torch.float32, device=dout.device)
        L.call('u3d_segment_gather_sum', L.ptr(dout), L.ptr(plan.vox_sp), L.ptr(plan.vox_offsets), plan.n_vox, C, 0,
               L.ptr(plan.sp_offsets), L.ptr(df), L.stream())
        return df, None


def superpoint_pool(feats: torch.Tensor, plan: PoolPlan) -> torch.Tensor:
    return _PoolFn.apply(feats, plan)


def superpoint_centers(points: torch.Tensor, sp_offsets: torch.Tensor, sp_points: torch.Tensor, S: int,
                       stats: Optional[torch.Tensor] = None, pt_offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mean over each superpoint's points of (xyz - scene_min); ``stats`` = VoxelBatch.stats
    (row b starts with the scene's min xyz) or None for raw coordinates (predict path)."""
    out = torch.empty(S, 3, dtype=torch.float32, device=points.device)
    B = 0 if stats is None else stats.shape[0]
    L.call('u3d_segment_mean_xyz', L.ptr(points), points.stride(0), L.ptr(sp_points), L.ptr(sp_offsets), S,
           L.ptr(stats), 12, L.ptr(pt_offsets), B, L.ptr(out), L.stream())
    return out


def instance_boxes(vb: VoxelBatch, instance_ids: torch.Tensor, n_inst_total: int, voxel_size: Optional[float] = None) -> torch.Tensor:
    """[n_inst_total, 6] = (centre xyz, size xyz) of the axis-aligned box around each instance's points in the
    scene-shifted frame; ``instance_ids`` int64 [Np] are batch-global (-1 = no instance).  The frame is
    ``xyz - scene min`` (unidet3d.py:300-301) or, when the batch carries elastic coordinates,
    ``(elastic - scene min) * voxel_size`` (:296-297; scaling by a positive constant commutes with min / max, so it is
    applied to the result).  One pass over the batch instead of the reference's per-instance boolean masks (:220-256)."""
    dev = vb.points.device
    src = vb.points if vb.coord_src is None else vb.coord_src
    mm = torch.empty(n_inst_total, 6, dtype=torch.float32, device=dev)
    ws = L.scratch(n_inst_total * 24 + 64, dev)
    L.call('u3d_segment_minmax_xyz', L.ptr(src), src.stride(0), L.ptr(instance_ids.contiguous()), src.shape[0],
           n_inst_total, L.ptr(vb.stats), 12, L.ptr(vb.pt_offsets), vb.stats.shape[0], L.ptr(mm), L.ptr(ws), L.stream())
    if vb.coord_src is not None:
        mm = mm * float(voxel_size)
    lo, hi = mm[:, :3], mm[:, 3:]
    return torch.cat(((hi + lo) / 2, hi - lo), 1)


# ----------------------------------------------------------------------------------------
# inference post-processing
# ----------------------------------------------------------------------------------------
NMS_MAX, NMS_ROT_MAX = 4400, 3600          # csrc/postproc.hip: boxes one workgroup keeps in LDS


def nms_multiclass(bboxes: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor, iou_thr: float, score_thr: float,
                   fast_nms: bool = True):
    """``UniDet3D._single_scene_multiclass_nms`` (unidet3d/unidet3d.py:595-650): class by class
    (ascending id), boxes above ``score_thr`` visited by descending score, greedy suppression by the BEV IoU
    (``fast_nms``: mmcv ``nms3d_normal``) or by the 3-D IoU of the corner boxes (mmdet3d ``aligned_3d_nms`` on
    ``_bbox_to_loss(boxes)``); 7-dof boxes go through mmcv ``nms3d`` (rotated BEV IoU) whatever ``fast_nms`` says.
    ``scores`` must already be sorted descending (they come from a sorted top-k).
    Returns (bboxes, scores, labels) in the reference's output order."""
    if bboxes.shape[1] not in (6, 7):
        raise ValueError('boxes must be (cx, cy, cz, dx, dy, dz[, heading])')
    sel = scores > score_thr
    bboxes, scores, labels = bboxes[sel], scores[sel], labels[sel]
    n = bboxes.shape[0]
    if n == 0:
        return bboxes.new_zeros((0, bboxes.shape[1])), bboxes.new_zeros((0,)), bboxes.new_zeros((0,))      # unidet3d.py:645-648: all three from `bboxes`
    order = torch.sort(labels, stable=True).indices            # (label asc, score desc)
    b = bboxes[order].contiguous().float()
    lab = labels[order].to(torch.int32).contiguous()
    keep = torch.empty(n, dtype=torch.uint8, device=b.device)
    if b.shape[1] == 7:                                           # with_yaw: mmcv nms3d on the rotated BEV rectangles (:625-626)
        fn, arg, limit = 'u3d_nms_rotated', b, NMS_ROT_MAX
    elif fast_nms:
        fn, arg, limit = 'u3d_nms_bev', b, NMS_MAX
    else:
        half = b[:, 3:] / 2                                      # _bbox_to_loss (criterion.py:180-198)
        fn, arg, limit = 'u3d_nms_aligned3d', torch.cat((b[:, :3] - half, b[:, :3] + half), dim=1).contiguous(), NMS_MAX
    # One workgroup holds a launch's boxes in LDS (4400 / 3600 boxes).  Suppression never crosses a class boundary, so a larger
    # set (test_cfg.topk_insts above the limit -- the reference accepts any) is cut at class boundaries into several launches;
    # only a SINGLE class with more boxes than the limit is refused.
    if n <= limit:
        L.call(fn, L.ptr(arg), L.ptr(lab), n, float(iou_thr), L.ptr(keep), L.stream())
    else:
        starts = torch.cat((lab.new_zeros(1), (lab[1:] != lab[:-1]).nonzero()[:, 0].to(torch.int32) + 1, lab.new_full((1,), n))).tolist()
        lo = 0
        while lo < n:
            hi = max([e for e in starts if lo < e <= lo + limit], default=None)
            if hi is None:
                raise L.U3DError(f'nms: one class holds more than {limit} boxes above the score threshold')
            L.call(fn, L.ptr(arg[lo:hi]), L.ptr(lab[lo:hi]), hi - lo, float(iou_thr), L.ptr(keep[lo:hi]), L.stream())
            lo = hi
    k = order[keep.bool()]
    out = bboxes[k]
    if fast_nms and out.shape[1] == 6:        # the reference appends a zero heading for nms3d_normal and returns those 7-column boxes (:629-638)
        out = torch.cat((out, torch.zeros_like(out[:, :1])), dim=1)
    return out, scores[k], labels[k]


def nms_bev_multiclass(bboxes, scores, labels, iou_thr, score_thr):
    return nms_multiclass(bboxes, scores, labels, iou_thr, score_thr, True)


def trim_boxes_by_superpoints(points: torch.Tensor, sp_offsets: torch.Tensor, sp_points: torch.Tensor, n_superpoints: int,
                              bboxes: torch.Tensor, low_sp_thr: float, up_sp_thr: float) -> torch.Tensor:
    """``UniDet3D.trim_bboxes_by_superpoints`` (unidet3d/unidet3d.py:540-593): [n,6] (centre, size) of the axis-aligned box
    around the points each box keeps after whole superpoints were deleted (< low) / added (> up).  ``bboxes`` [n,6] or [n,7]
    (heading: the inside test rotates the point shift by -yaw, ``get_face_distances`` :652-677).  ``(sp_offsets, sp_points)``
    is the CSR of point rows per superpoint (``csr_build`` / ``PoolPlan``)."""
    nb = bboxes.shape[0]
    mm = torch.empty(nb, 6, dtype=torch.float32, device=points.device)
    if nb:
        if bboxes.shape[1] == 7 and not bool((bboxes[:, 6] != 0).any()):
            bboxes = bboxes[:, :6]            # zero heading appended by the fast-NMS branch: the exact yaw-free path
        b = bboxes.contiguous().float()
        L.call('u3d_trim_boxes', L.ptr(points), points.stride(0), L.ptr(sp_points), L.ptr(sp_offsets), int(n_superpoints),
               L.ptr(b), nb, int(b.shape[1]), float(low_sp_thr), float(up_sp_thr), L.ptr(mm), L.stream())
    mn, mx = mm[:, :3], mm[:, 3:]
    return torch.cat(((mx + mn) / 2, mx - mn), dim=1)


def cat_views(ts):
    """``torch.cat(ts)`` -- without the copy (and without the slice / cat nodes in the autograd graph) when ``ts`` are the consecutive row
    slices of ONE contiguous tensor that together cover it, which is how the per-scene lists of the reference's interface are made
    here (``pooled[o_i:o_{i+1}]``): the base tensor itself is returned.  Anything else is concatenated."""
    ts = list(ts)
    if len(ts) == 1:
        return ts[0]
    base = getattr(ts[0], '_base', None)
    if base is not None and base.is_contiguous() and base.dim() >= 1:
        off, ok = base.storage_offset(), True
        for t in ts:
            # (also the autograd state and dtype of the base: slices taken under no_grad of a differentiable base must not hand the
            # base -- and with it a gradient path torch.cat(ts) would not have -- to the caller)
            if t._base is not base or not t.is_contiguous() or t.dim() != base.dim() or t.shape[1:] != base.shape[1:] or t.storage_offset() != off \
                    or t.requires_grad != base.requires_grad or t.dtype != base.dtype:
                ok = False
                break
            off += t.numel()
        if ok and off == base.storage_offset() + base.numel():
            return base
    return torch.cat(ts)


def offset_ids(ids, biases, keep_negative: bool = False) -> torch.Tensor:
    """``torch.cat([ids[i] + biases[i] for i])`` for per-scene int64 id tensors (superpoint / instance ids made batch-global) in a handful
    of multi-tensor launches whatever the number of scenes.  ``keep_negative``: ids < 0 ("no instance") stay as they are:
    id + bias * min(max(id + 1, 0), 1)."""
    ids, biases = list(ids), [int(b) for b in biases]
    if keep_negative:
        live = torch._foreach_add(ids, 1)
        torch._foreach_clamp_min_(live, 0)
        torch._foreach_clamp_max_(live, 1)
        torch._foreach_mul_(live, biases)
        out = torch._foreach_add(live, ids)
    else:
        out = torch._foreach_add(ids, biases)
    return torch.cat(out) if len(out) > 1 else out[0]

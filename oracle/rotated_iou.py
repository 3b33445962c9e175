"""CPU ORACLE (test infrastructure, NOT product code) -- differentiable DIoU of rotated 3-D boxes.

Prepared for SURVEY.md 8f rank 2 (ARKitScenes boxes of the joint config); the product raises NotImplementedError
for 7-dof boxes until the HIP kernels exist.  Only ``tests/`` may import this module.

Restates ``diff_diou_rotated_3d`` (unidet3d/rotated_iou_loss.py:14-60) in torch-CPU, fp32/fp64, differentiable by
autograd like the reference.  The BEV polygon intersection it calls lives in mmcv (@780ffed, Dockerfile:22-24:
``mmcv/ops/diff_iou_rotated.py`` -- ``box2corners``, ``box_intersection``, ``box1_in_box2``, ``build_vertices``,
``sort_indices`` (CUDA ``diff_iou_rotated_sort_vertices_forward``), ``calculate_area``), which is not under
/root/reference: PARITY UNPINNED.  Its published algorithm (lilanxiao/Rotated_IoU) is restated here stage by stage:
  1. corners of both rectangles (counter-clockwise from the +x+y corner, rotated by alpha);
  2. the 16 edge-edge intersection points with the parametric test 0 < t < 1 and 0 < u < 1;
  3. corners of one box inside the other (projection test with a 1e-6 slack);
  4. the <= 24 candidate vertices are sorted by angle around their mean and the polygon area is the shoelace sum.
Step 4's sort is done with atan2 here instead of mmcv's comparison kernel: the polygon and therefore the area are the
same (the sort only fixes the visiting order; gradients flow through the vertex coordinates, not the order).
mmcv's handling of exactly coincident boxes (duplicate vertices) is mirrored by dropping repeated vertices.
The oracle itself is pinned against first principles in tests/test_rotated_iou_cpu.py: closed-form overlaps,
Sutherland-Hodgman clipping written independently, rotation invariance, and the axis-aligned DIoU for alpha = 0.
"""
from __future__ import annotations

import torch


def box2corners(box: torch.Tensor) -> torch.Tensor:
    """(..., 5) (x, y, w, h, alpha) -> (..., 4, 2) corners, counter-clockwise starting at (+w/2, +h/2)."""
    x, y, w, h, a = box.unbind(-1)
    x4 = torch.stack([0.5 * w, -0.5 * w, -0.5 * w, 0.5 * w], -1)
    y4 = torch.stack([0.5 * h, 0.5 * h, -0.5 * h, -0.5 * h], -1)
    c, s = torch.cos(a)[..., None], torch.sin(a)[..., None]
    return torch.stack([x4 * c - y4 * s + x[..., None], x4 * s + y4 * c + y[..., None]], -1)


def _edge_intersections(c1: torch.Tensor, c2: torch.Tensor):
    """(..., 4, 2) x2 -> points (..., 16, 2), mask (..., 16)."""
    a1, a2 = c1[..., :, None, :], c1[..., [1, 2, 3, 0], :][..., :, None, :]          # edge i of box 1
    b1, b2 = c2[..., None, :, :], c2[..., [1, 2, 3, 0], :][..., None, :, :]          # edge j of box 2
    x1, y1, x2, y2 = a1[..., 0], a1[..., 1], a2[..., 0], a2[..., 1]
    x3, y3, x4, y4 = b1[..., 0], b1[..., 1], b2[..., 0], b2[..., 1]
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)
    par = num == 0
    safe = torch.where(par, torch.ones_like(num), num)
    t = torch.where(par, -torch.ones_like(num), den_t / safe)
    u = torch.where(par, -torch.ones_like(num), -den_u / safe)
    mask = (t > 0) & (t < 1) & (u > 0) & (u < 1)
    t2 = den_t / (num + 1e-8)
    pts = torch.stack([x1 + t2 * (x2 - x1), y1 + t2 * (y2 - y1)], -1) * mask[..., None].to(c1.dtype)
    return pts.flatten(-3, -2), mask.flatten(-2, -1)


def _inside(c1: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """corners of box 1 inside box 2: (..., 4) bool."""
    a, b, d = c2[..., 0:1, :], c2[..., 1:2, :], c2[..., 3:4, :]
    ab, ad, am = b - a, d - a, c1 - a
    pab, nab = (ab * am).sum(-1), (ab * ab).sum(-1)
    pad, nad = (ad * am).sum(-1), (ad * ad).sum(-1)
    return (pab / nab > -1e-6) & (pab / nab < 1 + 1e-6) & (pad / nad > -1e-6) & (pad / nad < 1 + 1e-6)


def oriented_box_intersection_2d(c1: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """Area of the intersection polygon of two rectangles given by corners (..., 4, 2)."""
    pts, m_int = _edge_intersections(c1, c2)
    verts = torch.cat([c1, c2, pts], -2)                                  # (..., 24, 2)
    mask = torch.cat([_inside(c1, c2), _inside(c2, c1), m_int], -1)       # (..., 24)
    shape = verts.shape[:-2]
    V, M = verts.reshape(-1, 24, 2), mask.reshape(-1, 24)
    areas = []
    for v, m in zip(V, M):
        p = v[m]
        if p.shape[0] >= 3:                                               # drop repeated vertices (coincident boxes / shared corners)
            keep = [0]
            for i in range(1, p.shape[0]):
                if all(float((p[i] - p[j]).detach().abs().max()) > 1e-8 for j in keep):
                    keep.append(i)
            p = p[keep]
        if p.shape[0] < 3:
            areas.append(v.sum() * 0)
            continue
        ctr = p.detach().mean(0)
        order = torch.argsort(torch.atan2(p[:, 1].detach() - ctr[1], p[:, 0].detach() - ctr[0]))
        q = p[order]
        r = torch.roll(q, -1, 0)
        areas.append((q[:, 0] * r[:, 1] - q[:, 1] * r[:, 0]).sum().abs() / 2)
    return torch.stack(areas).reshape(shape)


def diff_diou_rotated_3d(box3d1: torch.Tensor, box3d2: torch.Tensor) -> torch.Tensor:
    """unidet3d/rotated_iou_loss.py:14-60: (B, N, 7) (x, y, z, w, h, l, alpha) x2 -> (B, N) DIoU.

    Kept as the reference computes it, including its centre term: ``r2`` is taken over the first three entries of the
    BEV 5-vectors, i.e. (dx, dy, dw) -- not (dx, dy, dz) (:58)."""
    box1, box2 = box3d1[..., [0, 1, 3, 4, 6]], box3d2[..., [0, 1, 3, 4, 6]]
    c1, c2 = box2corners(box1), box2corners(box2)
    inter = oriented_box_intersection_2d(c1, c2)
    zmax1, zmin1 = box3d1[..., 2] + box3d1[..., 5] * 0.5, box3d1[..., 2] - box3d1[..., 5] * 0.5
    zmax2, zmin2 = box3d2[..., 2] + box3d2[..., 5] * 0.5, box3d2[..., 2] - box3d2[..., 5] * 0.5
    z_overlap = (torch.min(zmax1, zmax2) - torch.max(zmin1, zmin2)).clamp(min=0.)
    inter3d = inter * z_overlap
    vol1 = box3d1[..., 3] * box3d1[..., 4] * box3d1[..., 5]
    vol2 = box3d2[..., 3] * box3d2[..., 4] * box3d2[..., 5]
    union3d = vol1 + vol2 - inter3d
    x_max = torch.max(c1[..., 0].max(-1)[0], c2[..., 0].max(-1)[0]); x_min = torch.min(c1[..., 0].min(-1)[0], c2[..., 0].min(-1)[0])
    y_max = torch.max(c1[..., 1].max(-1)[0], c2[..., 1].max(-1)[0]); y_min = torch.min(c1[..., 1].min(-1)[0], c2[..., 1].min(-1)[0])
    z_max, z_min = torch.max(zmax1, zmax2), torch.min(zmin1, zmin2)
    r2 = ((box1[..., :3] - box2[..., :3]) ** 2).sum(-1)
    c2_ = (x_min - x_max) ** 2 + (y_min - y_max) ** 2 + (z_min - z_max) ** 2
    return inter3d / union3d - r2 / c2_


def rotated_diou_3d_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """rotated_iou_loss.py:63-82 without the mmdet ``weighted_loss`` wrapper: [N, 7] x2 -> [N] (1 - DIoU)."""
    return 1 - diff_diou_rotated_3d(pred[None], target[None])[0]

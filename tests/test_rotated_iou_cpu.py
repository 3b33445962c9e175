"""oracle/rotated_iou.py against first principles (CPU): closed forms, an independent Sutherland-Hodgman clipper,
invariances, the axis-aligned DIoU of the hot path for alpha = 0, and autograd vs finite differences."""
import math

import numpy as np
import torch

from oracle import criterion as oc
from oracle import rotated_iou as ri


def _clip_area(c1: np.ndarray, c2: np.ndarray) -> float:
    """Sutherland-Hodgman: clip polygon c1 by the convex polygon c2 (both counter-clockwise), shoelace area."""
    poly = [tuple(p) for p in c1]
    for i in range(len(c2)):
        a, b = c2[i], c2[(i + 1) % len(c2)]
        def inside(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0
        def cross_pt(p, q):
            dx, dy = q[0] - p[0], q[1] - p[1]
            ex, ey = b[0] - a[0], b[1] - a[1]
            t = ((a[0] - p[0]) * ey - (a[1] - p[1]) * ex) / (dx * ey - dy * ex)
            return (p[0] + t * dx, p[1] + t * dy)
        out = []
        for j in range(len(poly)):
            p, q = poly[j], poly[(j + 1) % len(poly)]
            if inside(q):
                if not inside(p):
                    out.append(cross_pt(p, q))
                out.append(q)
            elif inside(p):
                out.append(cross_pt(p, q))
        poly = out
        if not poly:
            return 0.0
    x = np.array([p[0] for p in poly]); y = np.array([p[1] for p in poly])
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def test_closed_form_overlaps():
    b = lambda *v: torch.tensor([v], dtype=torch.float64)
    area = lambda p, q: float(ri.oriented_box_intersection_2d(ri.box2corners(p), ri.box2corners(q)))
    assert abs(area(b(0, 0, 2, 2, 0), b(1, 1, 2, 2, 0)) - 1.0) < 1e-7                      # axis-aligned quarter overlap (mmcv's +1e-8 in the line parameter)
    assert abs(area(b(0, 0, 2, 2, 0), b(0, 0, 2, 2, math.pi / 4)) - 8 * (math.sqrt(2) - 1)) < 1e-7   # square vs itself at 45 deg: octagon
    assert abs(area(b(0, 0, 4, 4, 0), b(0.3, -0.2, 1, 2, 0.7)) - 2.0) < 1e-7              # fully contained
    assert area(b(0, 0, 1, 1, 0.3), b(5, 5, 1, 1, 1.1)) == 0.0                             # disjoint
    assert abs(area(b(0, 0, 2, 1, 0.4), b(0, 0, 2, 1, 0.4)) - 2.0) < 1e-7                  # identical boxes


def test_random_pairs_match_independent_clipper_and_invariances():
    g = torch.Generator().manual_seed(3)
    n = 200
    b1 = torch.cat([torch.randn(n, 2, generator=g), torch.rand(n, 2, generator=g) * 2 + 0.3, (torch.rand(n, 1, generator=g) - 0.5) * 6], 1).double()
    b2 = torch.cat([b1[:, :2] + torch.randn(n, 2, generator=g) * 0.7, torch.rand(n, 2, generator=g) * 2 + 0.3, (torch.rand(n, 1, generator=g) - 0.5) * 6], 1).double()
    c1, c2 = ri.box2corners(b1), ri.box2corners(b2)
    got = ri.oriented_box_intersection_2d(c1, c2).numpy()
    want = np.array([_clip_area(c1[i].numpy(), c2[i].numpy()) for i in range(n)])
    assert np.abs(got - want).max() < 1e-7 and (want > 0).sum() > 50
    assert np.abs(ri.oriented_box_intersection_2d(c2, c1).numpy() - got).max() < 1e-7           # symmetric
    th = 0.83                                                                                     # rotate both boxes about the origin
    R = torch.tensor([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]], dtype=torch.float64)
    r1 = torch.cat([b1[:, :2] @ R.T, b1[:, 2:4], b1[:, 4:] + th], 1); r2 = torch.cat([b2[:, :2] @ R.T, b2[:, 2:4], b2[:, 4:] + th], 1)
    assert np.abs(ri.oriented_box_intersection_2d(ri.box2corners(r1), ri.box2corners(r2)).numpy() - got).max() < 1e-7


def test_alpha_zero_iou_term_equals_axis_aligned_iou():
    g = torch.Generator().manual_seed(5)
    n = 64
    p = torch.cat([torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g) + 0.4, torch.zeros(n, 1)], 1).double()
    t = torch.cat([p[:, :3] + torch.randn(n, 3, generator=g) * 0.3, torch.rand(n, 3, generator=g) + 0.4, torch.zeros(n, 1)], 1).double()
    diou = ri.diff_diou_rotated_3d(p[None], t[None])[0]
    iou = oc.aligned_iou_3d(oc.bbox_to_loss(p[:, :6]), oc.bbox_to_loss(t[:, :6]))
    # the reference's rotated centre term is (dx, dy, dw)^2 / c2 (rotated_iou_loss.py:58); add it back to isolate the IoU
    c1, c2 = oc.bbox_to_loss(p[:, :6]), oc.bbox_to_loss(t[:, :6])
    cdiag = ((torch.minimum(c1[:, :3], c2[:, :3]) - torch.maximum(c1[:, 3:], c2[:, 3:])) ** 2).sum(-1)
    r2 = ((p[:, [0, 1, 3]] - t[:, [0, 1, 3]]) ** 2).sum(-1)
    assert torch.allclose(diou + r2 / cdiag, iou, atol=1e-7)


def test_gradients_match_finite_differences():
    p = torch.tensor([[0.1, -0.2, 0.3, 1.5, 1.0, 0.8, 0.4], [1.0, 1.0, 0.0, 2.0, 0.7, 1.2, -0.9]], dtype=torch.float64, requires_grad=True)
    t = torch.tensor([[0.4, 0.1, 0.2, 1.2, 1.3, 1.0, -0.3], [1.3, 0.6, 0.1, 1.1, 1.6, 0.9, 0.5]], dtype=torch.float64)
    loss = ri.rotated_diou_3d_loss(p, t).sum()
    loss.backward()
    num = torch.zeros_like(p)
    h = 1e-6
    with torch.no_grad():
        for i in range(p.shape[0]):
            for j in range(7):
                d = torch.zeros_like(p); d[i, j] = h
                num[i, j] = (ri.rotated_diou_3d_loss(p + d, t).sum() - ri.rotated_diou_3d_loss(p - d, t).sum()) / (2 * h)
    assert torch.allclose(p.grad, num, atol=1e-6), (p.grad, num)


def test_product_loss_matches_oracle_values_and_gradients():
    """unidet3d_amd.criterion.UniDet3DRotatedIoU3DLoss (batched tensor ops) against the oracle, incl. the [n, n_gt, 7] cost
    form, coincident / disjoint / contained boxes, weights and reductions."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.criterion import UniDet3DRotatedIoU3DLoss, diff_iou_rotated_3d
    g = torch.Generator().manual_seed(11)
    n = 96
    p = torch.cat([torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g) * 1.5 + 0.3, (torch.rand(n, 1, generator=g) - 0.5) * 6], 1).double()
    t = torch.cat([p[:, :3] + torch.randn(n, 3, generator=g) * 0.5, torch.rand(n, 3, generator=g) * 1.5 + 0.3, (torch.rand(n, 1, generator=g) - 0.5) * 6], 1).double()
    t[0] = p[0]                                   # identical boxes
    t[1, :2] += 50                                # disjoint
    t[2] = p[2]; t[2, 3:6] *= 0.3                 # contained, same centre and angle
    pp, po = p.clone().requires_grad_(), p.clone().requires_grad_()
    want = ri.rotated_diou_3d_loss(po, t)
    got = UniDet3DRotatedIoU3DLoss(mode='diou', reduction='none')(pp, t)
    assert torch.allclose(got, want, atol=1e-9)
    want.sum().backward(); got.sum().backward()
    assert torch.allclose(pp.grad[3:], po.grad[3:], atol=1e-7)          # generic pairs (the first three are non-smooth points)
    # cost-matrix form
    cm = UniDet3DRotatedIoU3DLoss(mode='diou', reduction='none')(p[:8, None].expand(8, 5, 7), t[None, :5].expand(8, 5, 7))
    ref = torch.stack([ri.rotated_diou_3d_loss(p[:8], t[j:j + 1].expand(8, 7)) for j in range(5)], 1)
    assert cm.shape == (8, 5) and torch.allclose(cm, ref, atol=1e-9)
    # iou mode, weights, reductions
    iou = diff_iou_rotated_3d(p, t, False)
    assert abs(float(iou[0]) - 1.0) < 1e-6 and float(iou[1]) == 0.0 and abs(float(iou[2]) - 0.027) < 1e-6
    w = torch.rand(n, generator=g).double()
    m = UniDet3DRotatedIoU3DLoss(mode='iou', reduction='mean', loss_weight=2.0)
    assert torch.allclose(m(p, t, weight=w), 2.0 * ((1 - iou) * w).mean())
    assert torch.allclose(m(p, t, weight=w, avg_factor=7.0), 2.0 * ((1 - iou) * w).sum() / (7.0 + torch.finfo(torch.float32).eps))
    assert float(m(p, t, weight=torch.zeros(n).double())) == 0.0
    # float32 (what the model runs in) stays within 1e-4 of the fp64 oracle
    got32 = UniDet3DRotatedIoU3DLoss(mode='diou', reduction='none')(p.float(), t.float())
    assert (got32.double() - want.detach()).abs().max() < 1e-4

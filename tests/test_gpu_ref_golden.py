"""Device-side repeat of tests/test_ref_golden_cpu.py on the MI355X: the criterion (packed and per-scene paths, incl. the
rotated DIoU and the rotated matcher branch of the joint config), ``get_targets`` / ``_select_queries`` and the superpoint
trimming kernel (yaw-free and rotated boxes) against vectors produced by the REAL reference files."""
import os

import numpy as np
import pytest
import torch

import test_ref_golden_cpu as R
from oracle import rotated_iou as orot

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
C, D, T = R.C, R.D, R.T


@pytest.mark.parametrize('tag,packed', [('C1', False), ('C1', True), ('C2', False), ('C3', False), ('C3', True)])
def test_criterion_on_device_matches_reference(tag, packed):
    worst = R.check_product_criterion(tag, DEV, packed)
    import _parity as PA
    PA.log_errors(f'criterion_golden_{tag}_{"packed" if packed else "loop"}', dict(grad_rel=worst))


def test_rotated_diou_values_and_gradients_on_device():
    """diff_diou_rotated_3d (unidet3d/rotated_iou_loss.py:14-60): values against the reference golden, gradients against the
    CPU oracle's autograd; plus the loss class with reduction='none' as the configs build it."""
    from unidet3d_amd import criterion as pc
    b1, b2 = T(C['F.rot_b1']), T(C['F.rot_b2'])
    p = b1.clone().to(DEV).requires_grad_()
    t = b2.clone().to(DEV)
    v = pc.diff_iou_rotated_3d(p, t, True)
    assert R.rel(v, C['F.rot_diou']) < 1e-4
    w = torch.linspace(0.5, 1.5, len(b1))
    (v * w.to(DEV)).sum().backward()
    po = b1.clone().requires_grad_()
    (orot.diff_diou_rotated_3d(po[None], b2[None])[0] * w).sum().backward()
    assert R.rel(p.grad, po.grad) < 1e-3
    assert float(p.grad[:, 6].abs().sum()) > 0                                     # the heading receives gradient
    loss = pc.UniDet3DRotatedIoU3DLoss(mode='diou', reduction='none')(p.detach(), t)
    assert R.rel(loss, 1 - C['F.rot_diou']) < 1e-4


def test_get_targets_and_select_queries_on_device():
    from unidet3d_amd.unidet3d import UniDet3D
    for tag in ('T0', 'T1', 'T2'):
        got = UniDet3D.get_targets(None, T(D[f'{tag}.pts']).to(DEV), R._boxes(D[f'{tag}.centers'], D[f'{tag}.sizes']).to(DEV), int(D[f'{tag}.topk']))
        assert torch.equal(got.cpu(), T(D[f'{tag}.targets']))
    R.select_queries_case(DEV)


@pytest.mark.parametrize('tag', ['P6', 'P7'])
def test_trim_kernel_matches_reference(tag):
    """u3d_trim_boxes against the reference's trim_bboxes_by_superpoints (inputs keep 1e-4 away from box faces and 0.02 from
    the ratio thresholds, so the min / max of the selected points is exact whatever the last bit of sin / cos)."""
    from unidet3d_amd import ops
    pts, sp, boxes = T(D[f'{tag}.pts']).to(DEV), T(D[f'{tag}.sp']).to(DEV), T(D[f'{tag}.boxes']).to(DEV)
    n_sp = int(sp.max()) + 1
    off, lst = ops.csr_build(sp, n_sp)
    got = ops.trim_boxes_by_superpoints(pts.contiguous(), off, lst, n_sp, boxes, 0.18, 0.81).cpu().numpy()
    want = np.concatenate((D[f'{tag}.centers'], D[f'{tag}.sizes']), 1)
    assert np.array_equal(got, want, equal_nan=True)


def test_instance_boxes_match_reference_get_bboxes_by_masks():
    """GT boxes from instance masks on the device (u3d_segment_minmax_xyz) == the reference's get_bboxes_by_masks."""
    from unidet3d_amd import ops
    masks, pts = T(D['B.masks']), T(D['B.pts'])
    ids = torch.where(masks.any(0), masks.float().argmax(0), -1)
    p6 = torch.cat((pts, torch.zeros(len(pts), 3)), 1).to(DEV)
    vb = ops.voxelize([p6], 0.05, 128)
    got = ops.instance_boxes(vb, ids.to(DEV), masks.shape[0])                     # in the frame xyz - scene min
    mn = pts.min(0)[0]
    want = torch.cat((T(D['B.centers']) - mn, T(D['B.sizes'])), 1)
    assert R.rel(got, want) < 1e-6

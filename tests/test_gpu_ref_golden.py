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


@pytest.mark.parametrize('tag,packed,fused', [('C1', False, False), ('C1', True, False), ('C1', True, True), ('C2', False, False), ('C2', True, True),
                                              ('C3', False, False), ('C3', True, False), ('C3', True, True)])
def test_criterion_on_device_matches_reference(tag, packed, fused):
    """per-scene loop, batched tensor ops and the fused kernel (csrc/criterion.hip) against the reference's values; C2 = a mixed
    batch of the joint config (three datasets, per-dataset top-k / weights, 7-dof ARKitScenes boxes -> rotated DIoU in matcher and
    loss): ('C2', True, True) runs it through the kernel's mixed-batch form"""
    worst = R.check_product_criterion(tag, DEV, packed, fused)
    import _parity as PA
    PA.log_errors(f'criterion_golden_{tag}_{"fused" if fused else ("packed" if packed else "loop")}', dict(grad_rel=worst))


@pytest.mark.parametrize('B,n_lo,n_hi,g_max,L', [(8, 1500, 2300, 12, 7), (3, 7, 40, 64, 2), (2, 3000, 3000, 1, 3)])
def test_fused_criterion_matches_tensor_op_path_at_bench_sizes(B, n_lo, n_hi, g_max, L):
    """csrc/criterion.hip vs the batched tensor-op formulation of the same arithmetic on random head outputs: loss and gradients
    (the matcher's discrete choices must coincide -- any difference shows up as an O(1e-3) loss change)."""
    from unidet3d_amd.structures import DepthInstance3DBoxes, InstanceData_
    from unidet3d_amd.registry import MODELS
    g = torch.Generator().manual_seed(B * 100 + L)
    sizes = [int(torch.randint(n_lo, n_hi + 1, (1,), generator=g)) for _ in range(B)]
    gts = [int(torch.randint(0, g_max + 1, (1,), generator=g)) for _ in range(B)]
    gts[0] = g_max
    if B > 2:
        gts[1] = 0                                                         # a scene without ground truth
    insts, cls, box = [], [[] for _ in range(L)], [[] for _ in range(L)]
    for n, k in zip(sizes, gts):
        gtb = torch.cat((torch.rand(k, 3, generator=g) * 3, torch.rand(k, 3, generator=g) + 0.2), 1)
        qm = torch.rand(k, n, generator=g) < 0.3
        if k:
            qm[0, 3:] = False                                              # a GT with fewer allowed queries than topk + 1
        insts.append(InstanceData_(labels_3d=torch.randint(0, 18, (k,), generator=g).to(DEV), query_masks=qm.to(DEV),
                                   bboxes_3d=DepthInstance3DBoxes(gtb, with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5)).to(DEV)))
        for l in range(L):
            cls[l].append(torch.randn(n, 19, generator=g) * 1.5)
            b = torch.cat((torch.rand(n, 3, generator=g) * 3, torch.rand(n, 3, generator=g) + 0.2), 1)
            if k:
                near = gtb[torch.randint(0, k, (n,), generator=g)] + torch.randn(n, 6, generator=g) * 0.08
                near[:, 3:] = near[:, 3:].abs() + 0.05
                b = torch.where((torch.rand(n, generator=g) < 0.5)[:, None], near, b)
            box[l].append(b)
    res = {}
    for fused in (True, False):
        pc_ = [torch.cat(c).to(DEV).requires_grad_() for c in cls]
        pb_ = [torch.cat(b).to(DEV).requires_grad_() for b in box]
        pred = dict(cls_preds=list(pc_[0].split(sizes)), bboxes=list(pb_[0].split(sizes)), aux_outputs=[None] * (L - 1),
                    _packed=dict(cls=pc_, box=pb_, sizes=sizes))
        crit = MODELS.build(R.SCANNET_CRIT)
        crit.fused = fused
        loss = crit(pred, insts, ['scannet'] * B)['det_loss']
        (loss * 1.7).backward()
        res[fused] = (loss.detach(), [t.grad for t in pc_], [t.grad for t in pb_])
    assert abs(float(res[True][0]) - float(res[False][0])) < 2e-6 * abs(float(res[False][0])), (res[True][0], res[False][0])
    for a, b in zip(res[True][1] + res[True][2], res[False][1] + res[False][2]):
        assert R.rel(a, b) < 2e-5


@pytest.mark.parametrize('seed', [0, 1])
def test_fused_criterion_mixed_batch_with_rotated_boxes_matches_the_per_scene_path(seed):
    """csrc/criterion.hip on a random MIXED batch (three datasets with different class counts, top-k and weights; one of them with
    7-dof boxes: rotated DIoU through dual numbers) vs the per-scene tensor-op path (criterion.get_layer_loss, pinned by the
    reference goldens C2 / F.rot): loss and every gradient, incl. the zero gradients in the columns of other datasets."""
    import _parity as PA
    from unidet3d_amd.structures import DepthInstance3DBoxes, InstanceData_
    from unidet3d_amd.registry import MODELS
    F = torch.nn.functional
    g = torch.Generator().manual_seed(100 + seed)
    crit_cfg = dict(R.JOINT_CRIT)
    names = ['scannet', 'arkitscenes', 's3dis', 'arkitscenes', 'scannet', 's3dis'][:5 + seed]
    n_cls = dict(scannet=18, s3dis=5, arkitscenes=17)
    L, B = 3, len(names)
    sizes = [int(torch.randint(40, 400, (1,), generator=g)) for _ in names]
    gts = [int(torch.randint(1, 9, (1,), generator=g)) for _ in names]
    gts[2] = 0                                                               # a scene without ground truth
    insts, cls, box = [], [[] for _ in range(L)], [[] for _ in range(L)]
    for name, n, k in zip(names, sizes, gts):
        yaw = name == 'arkitscenes'
        gtb = torch.cat((torch.rand(k, 3, generator=g) * 3, torch.rand(k, 3, generator=g) + 0.3), 1)
        if yaw:
            gtb = torch.cat((gtb, (torch.rand(k, 1, generator=g) - 0.5) * 2.4), 1)
        qm = torch.rand(k, n, generator=g) < 0.4
        insts.append(InstanceData_(labels_3d=torch.randint(0, n_cls[name], (k,), generator=g).to(DEV), query_masks=qm.to(DEV),
                                   bboxes_3d=DepthInstance3DBoxes(gtb, with_yaw=yaw, box_dim=7 if yaw else 6, origin=(0.5, 0.5, 0.5)).to(DEV)))
        for l in range(L):
            cls[l].append(torch.randn(n, n_cls[name] + 1, generator=g) * 1.5)
            b = torch.cat((torch.rand(n, 3, generator=g) * 3, torch.rand(n, 3, generator=g) + 0.3), 1)
            if yaw:
                b = torch.cat((b, (torch.rand(n, 1, generator=g) - 0.5) * 2.4), 1)
            if k:
                near = gtb[torch.randint(0, k, (n,), generator=g)] + torch.randn(n, gtb.shape[1], generator=g) * 0.08
                near[:, 3:6] = near[:, 3:6].abs() + 0.05
                b = torch.where((torch.rand(n, generator=g) < 0.5)[:, None], near, b)
            box[l].append(b)
    res = {}
    for fused in (True, False):
        lc = [[t.clone().to(DEV).requires_grad_() for t in c] for c in cls]
        lb = [[t.clone().to(DEV).requires_grad_() for t in b] for b in box]
        pred = dict(cls_preds=lc[0], bboxes=lb[0], aux_outputs=[dict(cls_preds=lc[l], bboxes=lb[l]) for l in range(1, L)])
        if fused:
            CU = max(n_cls.values()) + 1
            pred['_packed'] = dict(cls=[torch.cat([F.pad(t, (0, CU - t.shape[1])) for t in c]) for c in lc],
                                   box=[torch.cat([F.pad(t, (0, 7 - t.shape[1])) for t in b]) for b in lb], sizes=sizes,
                                   cidx=[list(range(n_cls[nm] + 1)) for nm in names], yaw=[nm == 'arkitscenes' for nm in names])
        crit = MODELS.build(crit_cfg)
        crit.fused = fused
        assert crit._can_fuse(pred, insts, names) == fused
        loss = crit(pred, insts, names)['det_loss']
        (loss * 1.3).backward()
        res[fused] = (float(loss.detach()), [t.grad for c in lc for t in c], [t.grad if t.grad is not None else torch.zeros_like(t) for b in lb for t in b])
    e_loss = abs(res[True][0] - res[False][0]) / abs(res[False][0])
    e_cls = max(R.rel(a, b) for a, b in zip(res[True][1], res[False][1]))
    e_box = max(R.rel(a, b) for a, b in zip(res[True][2], res[False][2]) if float(b.abs().max()) > 0)
    PA.log_errors(f'criterion_fused_mixed_rotated_seed{seed}', dict(loss_rel=e_loss, dcls=e_cls, dbox=e_box, loss=res[False][0]))
    print('fused mixed/rotated criterion vs per-scene path:', e_loss, e_cls, e_box)
    assert e_loss < 1e-5 and e_cls < 1e-4 and e_box < 1e-3


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

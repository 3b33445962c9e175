"""Host logic on CPU: the product criterion (torch ops, device agnostic) against the oracle's
line-by-line restatement of criterion.py, and the batched fast path against the per-scene loop."""
import torch

import unidet3d_amd  # noqa: F401
from oracle import criterion as oc
from unidet3d_amd.config import scannet_model_cfg
from unidet3d_amd.registry import MODELS
from unidet3d_amd.structures import DepthInstance3DBoxes, InstanceData_


def _case(sizes, n_gts, seed=0, layers=3):
    g = torch.Generator().manual_seed(seed)
    names = ['scannet'] * len(sizes)
    insts, oinsts = [], []
    for n, ng in zip(sizes, n_gts):
        c = torch.rand(ng, 3, generator=g) * 2
        s = torch.rand(ng, 3, generator=g) * 0.5 + 0.1
        labels = torch.randint(0, 18, (ng,), generator=g)
        qm = torch.rand(ng, n, generator=g) < 0.15
        b = DepthInstance3DBoxes(torch.cat((c, s), 1), box_dim=6)
        insts.append(InstanceData_(labels_3d=labels, bboxes_3d=b, query_masks=qm, sp_masks=qm))
        oinsts.append(oc.OInst(labels_3d=labels, bboxes_3d=oc.OBoxes(torch.cat((c, s), 1)), query_masks=qm))
    N = sum(sizes)
    packed_cls = [torch.randn(N, 19, generator=g).requires_grad_() for _ in range(layers)]
    packed_box = []
    for _ in range(layers):
        ctr = torch.rand(N, 3, generator=g) * 2
        sz = torch.rand(N, 3, generator=g) * 0.6 + 0.05
        packed_box.append(torch.cat((ctr, sz), 1).requires_grad_())

    def as_dict(packed: bool):
        outs = [dict(cls_preds=list(c.split(sizes)), bboxes=list(b.split(sizes))) for c, b in zip(packed_cls, packed_box)]
        d = dict(cls_preds=outs[0]['cls_preds'], bboxes=outs[0]['bboxes'], aux_outputs=outs[1:])
        if packed:
            d['_packed'] = dict(cls=packed_cls, box=packed_box, sizes=sizes)
        return d
    return names, insts, oinsts, as_dict, packed_cls, packed_box


def _crit():
    return MODELS.build(scannet_model_cfg()['criterion'])


def test_loop_path_matches_oracle_restatement():
    names, insts, oinsts, as_dict, *_ = _case([40, 25], [4, 3])
    loss = _crit()(as_dict(False), insts, names)['det_loss']
    ref = oc.criterion(as_dict(False), oinsts)
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))


def test_packed_path_equals_loop_path_values_and_grads():
    for sizes, n_gts, seed in (([50, 50, 50], [5, 2, 7], 1), ([30, 64, 9], [3, 0, 6], 2)):
        names, insts, _, as_dict, pc, pb = _case(sizes, n_gts, seed)
        crit = _crit()
        l_loop = crit(as_dict(False), insts, names)['det_loss']
        gl = torch.autograd.grad(l_loop, pc + pb)
        l_pack = crit(as_dict(True), insts, names)['det_loss']
        gp = torch.autograd.grad(l_pack, pc + pb)
        assert abs(float(l_loop) - float(l_pack)) < 1e-5 * abs(float(l_loop)), (float(l_loop), float(l_pack))
        for a, b in zip(gl, gp):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-4)


def test_decoder_box_decode_and_structures():
    b = DepthInstance3DBoxes(torch.tensor([[1.0, 2.0, 3.0, 0.5, 0.6, 0.8]]), box_dim=6)
    assert torch.allclose(b.tensor[0, 2], torch.tensor(3.0 - 0.4))
    assert torch.allclose(b.gravity_center[0], torch.tensor([1.0, 2.0, 3.0]))
    assert len(b[torch.tensor([0, 0])]) == 2


def test_rotated_boxes_take_the_rotated_loss_in_matcher_and_layer_loss():
    """7-dof ground truth (ARKitScenes, angles=True): the loop path matches by the rotated DIoU cost and sums the rotated DIoU
    loss (criterion.py:127-128, 264-265).  Checked against a hand evaluation with the rotated oracle on one scene / one layer."""
    from oracle import rotated_iou as ri
    g = torch.Generator().manual_seed(4)
    n, ng = 30, 3
    gt7 = torch.cat((torch.rand(ng, 3, generator=g) * 2, torch.rand(ng, 3, generator=g) * 0.6 + 0.2, (torch.rand(ng, 1, generator=g) - 0.5) * 3), 1)
    labels = torch.randint(0, 18, (ng,), generator=g)
    qm = torch.rand(ng, n, generator=g) < 0.3
    inst = InstanceData_(labels_3d=labels, bboxes_3d=DepthInstance3DBoxes(gt7, with_yaw=True, box_dim=7), query_masks=qm, sp_masks=qm)
    cls = torch.randn(n, 19, generator=g).requires_grad_()
    box = torch.cat((torch.rand(n, 3, generator=g) * 2, torch.rand(n, 3, generator=g) * 0.6 + 0.2, (torch.rand(n, 1, generator=g) - 0.5) * 3), 1).requires_grad_()
    crit = _crit()
    loss = crit.get_layer_loss(dict(cls_preds=[cls], bboxes=[box]), [inst], ['scannet'])
    # hand evaluation: cost = -0.5 * softmax(cls)[:, label] + 2.0 * (1 - DIoU); each GT keeps its 6 cheapest allowed queries
    gtb = torch.cat((inst.bboxes_3d.gravity_center, gt7[:, 3:]), 1)
    cost = -0.5 * cls.softmax(-1)[:, labels] + 2.0 * torch.stack([ri.rotated_diou_3d_loss(box, gtb[j:j + 1].expand(n, 7)) for j in range(ng)], 1)
    cost = torch.where(qm.T, cost, torch.tensor(1e8)).detach()
    kth = torch.topk(cost, 7, dim=0, largest=False).values[-1:]
    ids = torch.argwhere(cost < kth)
    target = torch.full((n,), 18); target[ids[:, 0]] = labels[ids[:, 1]]
    w = torch.ones(19); w[-1] = 0.1
    want = 0.5 * torch.nn.functional.cross_entropy(cls, target, w) + 1.0 * ri.rotated_diou_3d_loss(box[ids[:, 0]], gtb[ids[:, 1]]).mean()
    assert len(ids) > 0 and abs(float(loss) - float(want)) < 1e-5 * abs(float(want))
    loss.backward()
    assert torch.isfinite(box.grad).all() and box.grad[ids[:, 0]].abs().sum() > 0

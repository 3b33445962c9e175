"""The reference's 6-dataset joint config (BASELINE configs[3]) builds into the drop-in modules and its criterion runs on a
mixed batch that includes rotated (ARKitScenes) ground truth -- CPU, torch ops only (no kernel is launched: the model is only
constructed, the criterion is device agnostic).  Skipped where /root/reference is absent (the GPU box)."""
import os

import pytest
import torch

import unidet3d_amd  # noqa: F401
from unidet3d_amd.config import build_model, load_model_cfg
from unidet3d_amd.structures import DepthInstance3DBoxes, InstanceData_

CFG = '/root/reference/configs/unidet3d_1xb8_scannet_s3dis_multiscan_3rscan_scannetpp_arkitscenes.py'
pytestmark = pytest.mark.skipif(not os.path.exists(CFG), reason='reference configs are only available in the build container')


def test_packaged_joint_config_is_the_reference_fixture():
    """unidet3d_amd.config.joint_model_cfg() (what bench.py --config cfg4 builds) is the dict tools/gen_golden_reference.py wrote
    from the reference's joint config."""
    import json
    from unidet3d_amd.config import joint_model_cfg
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_joint_model_cfg.json')))
    assert joint_model_cfg() == ref


def test_joint_config_builds_and_flags_arrive():
    cfg = load_model_cfg(CFG)
    m = build_model(cfg)
    assert m.decoder.datasets == ['scannet', 's3dis', 'multiscan', '3rscan', 'scannetpp', 'arkitscenes']
    assert m.decoder.angles == [False] * 5 + [True] and m.fast_nms[1] is False and m.use_superpoints[3:] == [False] * 3
    assert m.criterion.topk == [6, 6, 3, 3, 3, 3] and m.test_cfg['iou_thr'][5] == 0.55
    n_cls = [len(i) for i in m.decoder.datasets_cls_idxs]
    assert n_cls[0] == 19 and all(k >= 2 for k in n_cls)              # per-dataset class columns + the shared no-object column
    assert m.get_dataset('data/arkitscenes/points/x.bin') == 'arkitscenes'


def test_joint_criterion_on_mixed_batch_with_rotated_ground_truth():
    cfg = load_model_cfg(CFG)
    m = build_model(cfg)
    g = torch.Generator().manual_seed(0)
    names = ['scannet', 'arkitscenes', 's3dis']
    dims = [6, 7, 6]
    sizes, n_gts = [20, 24, 16], [3, 4, 2]
    insts, cls, box = [], [], []
    for name, d, n, ng in zip(names, dims, sizes, n_gts):
        k = len(m.decoder.datasets_cls_idxs[m.decoder.datasets.index(name)])
        gt = torch.cat((torch.rand(ng, 3, generator=g) * 2, torch.rand(ng, 3, generator=g) * 0.6 + 0.2), 1)
        if d == 7:
            gt = torch.cat((gt, (torch.rand(ng, 1, generator=g) - 0.5) * 3), 1)
        qm = torch.rand(ng, n, generator=g) < 0.4
        insts.append(InstanceData_(labels_3d=torch.randint(0, k - 1, (ng,), generator=g), query_masks=qm, sp_masks=qm,
                                   bboxes_3d=DepthInstance3DBoxes(gt, with_yaw=d == 7, box_dim=d)))
        cls.append(torch.randn(n, k, generator=g).requires_grad_())
        b = torch.cat((torch.rand(n, 3, generator=g) * 2, torch.rand(n, 3, generator=g) * 0.6 + 0.2), 1)
        if d == 7:
            b = torch.cat((b, (torch.rand(n, 1, generator=g) - 0.5) * 3), 1)
        box.append(b.requires_grad_())
    out = dict(cls_preds=cls, bboxes=box, aux_outputs=[dict(cls_preds=cls, bboxes=box)])
    loss = m.criterion(out, insts, names)['det_loss']
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(b.grad).all() for b in box)
    assert box[1].grad[:, 6].abs().sum() > 0                            # the heading of the rotated boxes receives gradient


def test_prefetch_is_a_no_op_off_gpu_and_tensor_walk_finds_nested_tensors():
    """host logic of UniDet3D.prefetch: on CPU tensors nothing is staged (``loss`` then does the work inline); the walker that
    hands side-stream tensors to the main stream (record_stream) reaches tensors behind lists / dicts / object attributes."""
    import torch
    from unidet3d_amd.unidet3d import _tensors_of
    from unidet3d_amd.config import build_model, scannet_model_cfg

    class Box:
        def __init__(self, t):
            self.tensor = t
    t1, t2, t3 = torch.zeros(2), torch.ones(3), torch.arange(4)
    seen = list(_tensors_of(dict(a=[t1, (Box(t2),)], b=dict(c=t3), d=7, e='x', f=t1)))
    assert seen == []                                   # CPU tensors are not stream-managed: nothing to record

    class Slotted:                                      # no __dict__: attributes live in __slots__
        __slots__ = ('t', 'more')

        def __init__(self, t, more):
            self.t, self.more = t, more
    deep = t3
    for _ in range(40):                                 # far beyond any fixed depth cap
        deep = [dict(x=deep)]
    found = list(_tensors_of(dict(a=[t1, (Box(t2),)], s=Slotted(t1, Slotted(t2, None)), deep=deep, f=t1), cuda_only=False))
    assert len(found) == 3 and {id(t) for t in found} == {id(t1), id(t2), id(t3)}
    import pytest
    # an object the walk cannot look into (no __dict__ / __slots__, not a container) is an opaque leaf: one warning per type,
    # the tensors next to it are still found (a datetime / Enum / Path in a batch must not stop training)
    import datetime
    with pytest.warns(UserWarning, match='opaque leaf'):
        got = list(_tensors_of([object(), datetime.date(2024, 1, 1), t1], cuda_only=False))
    assert len(got) == 1 and got[0] is t1
    model = build_model(scannet_model_cfg(voxel_size=0.05))
    model.prefetch(dict(points=[torch.zeros(10, 6)]), [])
    assert model._prefetched is None
    model.prefetch_step([dict(inputs=dict(points=torch.zeros(10, 6)), data_samples=None)])
    assert model._staged is None

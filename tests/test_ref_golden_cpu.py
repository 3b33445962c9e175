"""Oracle AND product host logic against vectors produced by the REAL reference files (tools/gen_golden_reference.py,
run in the build container): criterion / matcher / target assignment (criterion.py), get_targets / _select_queries /
get_bboxes_by_masks / trim_bboxes_by_superpoints (unidet3d.py, AST-extracted), the module tree's state_dict keys
(spconv_unet.py, encoder.py), indoor_eval (indoor_eval.py) and the pipeline transforms (transforms_3d.py).
Everything here runs on CPU tensors (the criterion and the host transforms are device-agnostic torch / numpy);
tests/test_gpu_ref_golden.py repeats the device-side parts on the MI355X."""
import json
import os

import numpy as np
import pytest
import torch

import unidet3d_amd  # noqa: F401
from oracle import criterion as oc
from oracle import postproc as pp
from oracle import rotated_iou as orot
from unidet3d_amd import criterion as pc
from unidet3d_amd.registry import MODELS, TASK_UTILS
from unidet3d_amd.structures import DepthInstance3DBoxes, InstanceData_

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
C = np.load(os.path.join(GOLD, 'ref_criterion.npz'))
D = np.load(os.path.join(GOLD, 'ref_detector.npz'))


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12)) if a.numel() else 0.0


SCANNET_CRIT = dict(
    type='UniDet3DCriterion', datasets=['scannet'], datasets_weights=[1],
    bbox_loss_simple=dict(type='UniDet3DAxisAlignedIoULoss', mode='diou', reduction='none'),
    bbox_loss_rotated=dict(type='UniDet3DRotatedIoU3DLoss', mode='diou', reduction='none'),
    matcher=dict(type='UniMatcher', costs=[
        dict(type='QueryClassificationCost', weight=0.5),
        dict(type='BboxCostJointTraining', weight=2.0,
             loss_simple=dict(type='UniDet3DAxisAlignedIoULoss', mode='diou', reduction='none'),
             loss_rotated=dict(type='UniDet3DRotatedIoU3DLoss', mode='diou', reduction='none'))]),
    loss_weight=[0.5, 1.0], non_object_weight=0.1, topk=[6], iter_matcher=True)
JOINT_CRIT = dict(SCANNET_CRIT, datasets=['scannet', 's3dis', 'arkitscenes'], datasets_weights=[1, 0.7, 2.5], topk=[6, 4, 3])
CASES = {'C1': (SCANNET_CRIT, 3, 3), 'C2': (JOINT_CRIT, 4, 2), 'C3': (SCANNET_CRIT, 2, 2)}


def load_case(tag, device='cpu', packed=False):
    """-> (pred dict with leaf tensors requiring grad, insts, names, n_scenes, L)."""
    cfg, n_sc, L = CASES[tag]
    names = [str(s) for s in C[f'{tag}.names']]
    cls = [[T(C[f'{tag}.L{l}.cls{i}']).to(device).requires_grad_() for i in range(n_sc)] for l in range(L)]
    box = [[T(C[f'{tag}.L{l}.box{i}']).to(device).requires_grad_() for i in range(n_sc)] for l in range(L)]
    insts = []
    for i in range(n_sc):
        gtb = T(C[f'{tag}.s{i}.gtb']).to(device)
        dof = gtb.shape[1] if gtb.numel() else box[0][i].shape[1]
        insts.append(InstanceData_(labels_3d=T(C[f'{tag}.s{i}.labels']).to(device), query_masks=T(C[f'{tag}.s{i}.qm']).to(device),
                                   bboxes_3d=DepthInstance3DBoxes(gtb.reshape(-1, dof), with_yaw=dof == 7, box_dim=dof, origin=(0.5, 0.5, 0.5))))
    pred = dict(cls_preds=cls[0], bboxes=box[0], aux_outputs=[dict(cls_preds=cls[l], bboxes=box[l]) for l in range(1, L)])
    if packed and len(set(names)) == 1:       # what UniDet3DEncoder adds for a single-dataset batch: [sum n_i, .] matrices, final layer first
        pc_, pb_ = [torch.cat(c) for c in cls], [torch.cat(b) for b in box]
        pred['_packed'] = dict(cls=pc_, box=pb_, sizes=[int(t.shape[0]) for t in cls[0]])
    elif packed:     # ... and for a mixed batch: logits in a common width (a scene's classes first), boxes in 7 columns, plus which is which
        F = torch.nn.functional
        CU = max(int(t.shape[1]) for t in cls[0])
        pc_ = [torch.cat([F.pad(t, (0, CU - t.shape[1])) for t in c]) for c in cls]
        pb_ = [torch.cat([F.pad(t, (0, 7 - t.shape[1])) for t in b]) for b in box]
        pred['_packed'] = dict(cls=pc_, box=pb_, sizes=[int(t.shape[0]) for t in cls[0]], cidx=[list(range(int(t.shape[1]))) for t in cls[0]],
                               yaw=[int(t.shape[1]) == 7 for t in box[0]])
    return cfg, pred, insts, names, cls, box


# ------------------------------------------------------------------------------------------------ matcher / costs
def test_matcher_class_cost_only_matches_reference():
    """UniMatcher + QueryClassificationCost of the reference with NO stubbed arithmetic behind the golden (M0)."""
    scores, labels, qm = T(C['M0.scores']), T(C['M0.labels']), T(C['M0.qm'])
    m = TASK_UTILS.build(dict(type='UniMatcher', costs=[dict(type='QueryClassificationCost', weight=0.5)]))
    iq, ig = m(InstanceData_(scores=scores), InstanceData_(labels=labels, query_masks=qm), 6)
    assert torch.equal(iq, T(C['M0.iq'])) and torch.equal(ig, T(C['M0.ig']))
    cost = pc.QueryClassificationCost(0.5)(InstanceData_(scores=scores), InstanceData_(labels=labels))
    assert rel(cost, C['M0.cost']) < 1e-6
    # oracle: same matcher with the box cost switched off
    oq, og = oc.uni_matcher(scores, torch.zeros(37, 6), labels, torch.ones(6, 6), qm, 6, w_box=0.0)
    assert torch.equal(oq, T(C['M0.iq'])) and torch.equal(og, T(C['M0.ig']))


@pytest.mark.parametrize('tag', ['C1', 'C3'])
def test_oracle_criterion_matches_reference(tag):
    cfg, pred, insts, names, cls, box = load_case(tag)
    oinsts = [oc.OInst(labels_3d=i.labels_3d, query_masks=i.query_masks,
                       bboxes_3d=oc.OBoxes(torch.cat((i.bboxes_3d.gravity_center, i.bboxes_3d.tensor[:, 3:6]), 1))) for i in insts]
    loss = oc.criterion(pred, oinsts)
    assert abs(float(loss) - float(C[f'{tag}.loss'])) < 1e-6 * abs(float(C[f'{tag}.loss']))
    loss.backward()
    _, n_sc, L = CASES[tag]
    for l in range(L):
        for i in range(n_sc):
            iq, ig = oc.uni_matcher(cls[l][i].detach(), box[l][i].detach(), oinsts[i].labels_3d,
                                    torch.cat((oinsts[i].bboxes_3d.gravity_center, oinsts[i].bboxes_3d.tensor[:, 3:6]), 1),
                                    oinsts[i].query_masks, 6)
            assert torch.equal(iq, T(C[f'{tag}.L{l}.iq{i}'])) and torch.equal(ig, T(C[f'{tag}.L{l}.ig{i}']))
            assert rel(cls[l][i].grad, C[f'{tag}.L{l}.gcls{i}']) < 1e-5
            if box[l][i].grad is not None:
                assert rel(box[l][i].grad, C[f'{tag}.L{l}.gbox{i}']) < 1e-5


def test_oracle_joint_criterion_and_get_targets_match_reference():
    """Oracle with per-dataset weights / topk and the rotated DIoU (C2), and oracle get_targets (T*)."""
    cfg, pred, insts, names, cls, box = load_case('C2')
    oinsts = [oc.OInst(labels_3d=i.labels_3d, query_masks=i.query_masks,
                       bboxes_3d=oc.OBoxes(torch.cat((i.bboxes_3d.gravity_center, i.bboxes_3d.tensor[:, 3:]), 1), i.bboxes_3d.with_yaw))
              for i in insts]
    idx = [cfg['datasets'].index(n) for n in names]
    loss = oc.criterion(pred, oinsts, topk=[cfg['topk'][i] for i in idx], dataset_weight=[cfg['datasets_weights'][i] for i in idx])
    assert abs(float(loss.detach()) - float(C['C2.loss'])) < 2e-6 * abs(float(C['C2.loss']))
    loss.backward()
    for l in range(2):
        for i in range(4):
            assert rel(cls[l][i].grad, C[f'C2.L{l}.gcls{i}']) < 1e-5
            g = box[l][i].grad
            assert rel(g if g is not None else torch.zeros_like(box[l][i]), C[f'C2.L{l}.gbox{i}']) < 1e-4
    for tag in ('T0', 'T1', 'T2'):
        assert torch.equal(oc.get_targets(T(D[f'{tag}.pts']), T(D[f'{tag}.centers']), int(D[f'{tag}.topk'])), T(D[f'{tag}.targets']))


def check_product_criterion(tag, device, packed, fused=True):
    cfg, pred, insts, names, cls, box = load_case(tag, device, packed)
    crit = MODELS.build(cfg)
    crit.fused = fused          # on a GPU the packed single-dataset path runs csrc/criterion.hip unless told otherwise
    if packed:
        assert crit._can_pack(pred, insts, names) or (fused and crit._can_fuse(pred, insts, names))
    loss = crit(pred, insts, names)['det_loss']
    want = float(C[f'{tag}.loss'])
    assert abs(float(loss) - want) < 2e-6 * abs(want), (float(loss), want)
    loss.backward()
    _, n_sc, L = CASES[tag]
    worst = 0.0
    for l in range(L):
        for i in range(n_sc):
            worst = max(worst, rel(cls[l][i].grad, C[f'{tag}.L{l}.gcls{i}']))
            g = box[l][i].grad
            worst = max(worst, rel(g if g is not None else torch.zeros_like(box[l][i]), C[f'{tag}.L{l}.gbox{i}']))
    assert worst < 2e-5, worst
    # the matcher alone, layer by layer (indices are integers: exact)
    with torch.no_grad():
        for l in range(L):
            for i in range(n_sc):
                idx = crit.datasets.index(names[i])
                p = InstanceData_(scores=cls[l][i].detach(), bboxes=box[l][i].detach())
                gt = InstanceData_(labels=insts[i].labels_3d, query_masks=insts[i].query_masks, bboxes=pc._gt_boxes(insts[i].bboxes_3d))
                iq, ig = crit.matcher(p, gt, crit.topk[idx])
                assert torch.equal(iq.cpu(), T(C[f'{tag}.L{l}.iq{i}'])) and torch.equal(ig.cpu(), T(C[f'{tag}.L{l}.ig{i}'])), (tag, l, i)
    return worst


@pytest.mark.parametrize('tag,packed', [('C1', False), ('C1', True), ('C2', False), ('C3', False), ('C3', True)])
def test_product_criterion_matches_reference(tag, packed):
    check_product_criterion(tag, 'cpu', packed)


def test_layer_loss_and_pure_loss_functions():
    cfg, pred, insts, names, cls, box = load_case('C1')
    crit = MODELS.build(cfg)
    l0 = crit.get_layer_loss(dict(cls_preds=[t.detach() for t in cls[0]], bboxes=[t.detach() for t in box[0]]), insts, names)
    assert abs(float(l0) - float(C['C1.layer0_loss'])) < 1e-6 * abs(float(C['C1.layer0_loss']))
    p, t = T(C['F.diou_p']), T(C['F.diou_t'])
    for fn in (pc.axis_aligned_diou_loss, oc.axis_aligned_diou_loss):
        assert rel(fn(p, t), C['F.diou']) < 1e-6
        assert rel(fn(p[:, None].expand(50, 4, 6), t[None, :4].expand(50, 4, 6)), C['F.diou_matrix']) < 1e-6     # the [:, 0] quirk
    b1, b2 = T(C['F.rot_b1']), T(C['F.rot_b2'])
    assert rel(pc.diff_iou_rotated_3d(b1, b2, True), C['F.rot_diou']) < 1e-4
    assert rel(orot.diff_diou_rotated_3d(b1[None], b2[None])[0], C['F.rot_diou']) < 1e-5


# ------------------------------------------------------------------------------------------------ detector methods
def _boxes(centers, sizes):
    return DepthInstance3DBoxes(torch.cat((T(centers), T(sizes)), 1), with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5))


@pytest.mark.parametrize('tag', ['T0', 'T1', 'T2'])
def test_get_targets_matches_reference(tag):
    from unidet3d_amd.unidet3d import UniDet3D
    got = UniDet3D.get_targets(None, T(D[f'{tag}.pts']), _boxes(D[f'{tag}.centers'], D[f'{tag}.sizes']), int(D[f'{tag}.topk']))
    assert got.dtype == torch.bool and torch.equal(got, T(D[f'{tag}.targets']))


def select_queries_case(device):
    from unidet3d_amd.unidet3d import UniDet3D

    class S:
        query_thr = int(D['Q.query_thr'])
    x = [T(D[f'Q.x{i}']).to(device) for i in range(3)]
    insts = [InstanceData_(sp_centers=T(D[f'Q.centers{i}']).to(device), sp_masks=T(D[f'Q.sp_masks{i}']).to(device)) for i in range(3)]
    perms = [T(D[f'Q.perm{i}']) for i in range(3)]
    q, c, gi = UniDet3D._select_queries(S(), x, insts, perms)
    for i in range(3):
        assert torch.equal(q[i].cpu(), T(D[f'Q.q{i}'])) and torch.equal(c[i].cpu(), T(D[f'Q.c{i}']))
        assert torch.equal(gi[i].query_masks.cpu(), T(D[f'Q.qm{i}']))
        assert torch.equal(gi[i].sp_centers.cpu(), T(D[f'Q.c{i}']))


def test_select_queries_matches_reference():
    select_queries_case('cpu')


def test_get_bboxes_by_masks_matches_reference():
    from unidet3d_amd.unidet3d import UniDet3D
    masks, pts = T(D['B.masks']), T(D['B.pts'])
    ids = torch.where(masks.any(0), masks.float().argmax(0), -1)
    b = UniDet3D.get_bboxes_by_masks(ids, masks.shape[0], pts)
    assert torch.equal(b.gravity_center, T(D['B.centers'])) and torch.equal(b.tensor[:, 3:6], T(D['B.sizes']))


@pytest.mark.parametrize('tag', ['P6', 'P7'])
def test_oracle_trim_boxes_matches_reference(tag):
    """oracle/postproc.trim_boxes (incl. the rotated face test) against the reference's trim_bboxes_by_superpoints."""
    got = pp.trim_boxes(D[f'{tag}.pts'], D[f'{tag}.sp'], D[f'{tag}.boxes'], 0.18, 0.81)
    want = np.concatenate((D[f'{tag}.centers'], D[f'{tag}.sizes']), 1)
    assert np.isinf(want[-1]).any() or np.isnan(want[-1]).any()          # the box that holds no point
    assert np.array_equal(got, want, equal_nan=True)


# ------------------------------------------------------------------------------------------------ module tree / checkpoints
def test_state_dict_keys_and_strict_checkpoint_load():
    """The product's parameter / buffer names and shapes equal the reference's module tree; a checkpoint written with the
    reference's key list loads strict=True, round-trips, and a renamed key is rejected."""
    from _detw import det_array
    from unidet3d_amd.config import build_model, scannet_model_cfg
    tree = json.load(open(os.path.join(GOLD, 'ref_state_dict_keys.json')))
    model = build_model(scannet_model_cfg())
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(tree['keys'].keys())
    assert all(list(sd[k].shape) == tree['keys'][k] for k in sd)
    assert sum(p.numel() for p in model.parameters()) == tree['n_params'] == 15846427
    ckpt = {}
    for i, (k, shape) in enumerate(sorted(tree['keys'].items())):        # a "released checkpoint": reference keys, arbitrary values
        ckpt[k] = torch.tensor(7, dtype=torch.long) if k.endswith('num_batches_tracked') else torch.from_numpy(det_array(i, tuple(shape)).copy())
    res = model.load_state_dict({'state_dict': ckpt}['state_dict'], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    back = model.state_dict()
    assert all(torch.equal(back[k], ckpt[k]) for k in ckpt)
    bad = dict(ckpt)
    bad['unet.blocks.block0.conv_branch.2.weights'] = bad.pop('unet.blocks.block0.conv_branch.2.weight')
    with pytest.raises(RuntimeError):
        model.load_state_dict(bad, strict=True)


# ------------------------------------------------------------------------------------------------ evaluation
def test_indoor_eval_matches_reference():
    from unidet3d_amd.evaluation import average_precision, indoor_eval
    E = np.load(os.path.join(GOLD, 'ref_eval.npz'))
    gt, dt = [], []
    for i in range(int(E['E.n_img'])):
        gb, db = E[f'E.gt_box{i}'], E[f'E.dt_box{i}']
        mk = lambda b: DepthInstance3DBoxes(np.concatenate((b, np.zeros((len(b), 1), np.float32)), 1), with_yaw=False, box_dim=7, origin=(0.5, 0.5, 0.5))  # noqa: E731
        gt.append(dict(gt_bboxes_3d=mk(gb), gt_labels_3d=E[f'E.gt_lab{i}'].tolist()))
        dt.append(dict(labels_3d=T(E[f'E.dt_lab{i}']), scores_3d=T(E[f'E.dt_score{i}']), bboxes_3d=mk(db)))
    ret = indoor_eval(gt, dt, [0.25, 0.5], {i: f'cls{i}' for i in range(int(E['E.n_cls']))})
    keys = sorted(ret.keys())
    assert keys == [str(k) for k in E['E.ret_keys']]
    got = np.array([ret[k] for k in keys])
    assert np.allclose(got, E['E.ret_vals'], rtol=0, atol=1e-7, equal_nan=True)
    assert 0.0 < ret['mAP_0.50'] < ret['mAP_0.25'] < 1.0
    assert np.array_equal(average_precision(E['E.ap_rec'], E['E.ap_prec']), E['E.ap_area'])
    assert np.array_equal(average_precision(E['E.ap_rec'], E['E.ap_prec'], '11points'), E['E.ap_11'])


# ------------------------------------------------------------------------------------------------ transforms
def test_transforms_match_reference():
    from unidet3d_amd import transforms as X
    G = np.load(os.path.join(GOLD, 'ref_transforms.npz'))
    pts = G['X.points']
    np.random.seed(1234)
    e = X.ElasticTransfrom([6, 20], [40, 160], 0.02, 1.0)(dict(points=pts.copy()))['elastic_coords']
    assert e.dtype == G['X.elastic'].dtype and np.abs(e - G['X.elastic']).max() < 1e-4          # voxel units; distortion is ~50 voxels
    assert np.abs(G['X.elastic'] - pts[:, :3] / 0.02).max() > 10
    np.random.seed(1234)
    e0 = X.ElasticTransfrom([6, 20], [40, 160], 0.02, 0.0)(dict(points=torch.from_numpy(pts.copy())))['elastic_coords']
    assert e0.dtype == G['X.elastic_off'].dtype and np.array_equal(e0, G['X.elastic_off'])
    d = X.PointDetClassMappingScanNet(20, [0, 1])(dict(pts_instance_mask=G['X.sn.inst'].copy(), pts_semantic_mask=G['X.sn.sem'].copy(),
                                                        sp_pts_mask=G['X.sn.sp'].copy()))
    assert np.array_equal(d['pts_instance_mask'], G['X.sn.out_inst']) and np.array_equal(d['gt_labels_3d'], G['X.sn.out_labels'])
    assert np.array_equal(d['gt_sp_masks'].numpy(), G['X.sn.out_sp_masks'])
    d3 = X.PointDetClassMappingS3DIS([7, 8, 9, 10, 11])(dict(pts_instance_mask=G['X.s3.inst'].copy(), pts_semantic_mask=G['X.s3.sem'].copy(),
                                                            sp_pts_mask=G['X.sn.sp'].copy()))
    assert np.array_equal(d3['pts_instance_mask'], G['X.s3.out_inst']) and np.array_equal(d3['gt_labels_3d'].numpy(), G['X.s3.out_labels'])
    assert np.array_equal(d3['gt_sp_masks'].numpy(), G['X.s3.out_sp_masks'])
    np.random.seed(99)
    dd = X.PointSample_(1500)(dict(points=pts.copy(), pts_instance_mask=G['X.ps.in_inst'].copy(), pts_semantic_mask=G['X.sn.sem'].copy(),
                                   sp_pts_mask=G['X.sn.sp'].copy()))
    assert np.array_equal(dd['points'], G['X.ps.points']) and np.array_equal(dd['pts_instance_mask'], G['X.ps.inst'])
    assert np.array_equal(dd['pts_semantic_mask'], G['X.ps.sem']) and np.array_equal(dd['sp_pts_mask'], G['X.ps.sp'])
    n = X.NormalizePointsColor_(color_mean=[127.5, 127.5, 127.5])(dict(points=pts.copy()))['points']
    assert np.allclose(n[:, 3:], (pts[:, 3:] - 127.5) / 127.5) and np.array_equal(n[:, :3], pts[:, :3])


def test_on_disk_contract_round_trip(tmp_path):
    """points/*.bin float32 [N,6], super_points / instance / semantic *.bin int64 [N] -> transformed batch inputs."""
    from unidet3d_amd import transforms as X
    G = np.load(os.path.join(GOLD, 'ref_transforms.npz'))
    (tmp_path / 'scannet').mkdir()
    paths = {k: tmp_path / 'scannet' / f'{k}.bin' for k in ('points', 'sp', 'inst', 'sem')}
    G['X.points'].tofile(paths['points']); G['X.sn.sp'].astype(np.int64).tofile(paths['sp'])
    G['X.sn.inst'].astype(np.int64).tofile(paths['inst']); G['X.sn.sem'].astype(np.int64).tofile(paths['sem'])
    d = X.load_scene_bins(paths['points'], paths['sp'], paths['inst'], paths['sem'])
    for t in (X.PointSample_(2000), X.PointDetClassMappingScanNet(20, [0, 1]), X.NormalizePointsColor_([127.5] * 3),
              X.ElasticTransfrom([6, 20], [40, 160], 0.02)):
        d = t(d)
    inputs, samples = X.to_batch_inputs([d], 'cpu')
    assert inputs['points'][0].shape == (2000, 6) and inputs['elastic_coords'][0].shape == (2000, 3)
    assert samples[0].gt_instances_3d.sp_masks.shape == (len(d['gt_labels_3d']), samples[0].n_superpoints)
    assert 'scannet' in samples[0].lidar_path.split('/')

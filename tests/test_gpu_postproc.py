"""GPU parity of the inference post-processing (u3d_nms_bev, u3d_trim_boxes, UniDet3D.predict) against
oracle/postproc.py: bit-exact (integer / comparison work), including the empty and degenerate cases."""
import numpy as np
import pytest
import torch

from _detw import fill_state_dict
from oracle import postproc as pp

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
F32 = np.float32


def _boxes(rng, n, spread=4.0):
    c = rng.uniform(-spread, spread, (n, 3)); d = rng.uniform(0.2, 2.5, (n, 3))
    return np.concatenate([c, d], 1).astype(F32)


@pytest.mark.parametrize('fast', [True, False])
@pytest.mark.parametrize('n,n_cls,thr,score_thr', [(0, 3, 0.5, 0.0), (1, 1, 0.5, 0.0), (300, 5, 0.5, 0.0), (1000, 18, 0.5, 0.0),
                                                   (1000, 18, 0.25, 0.3), (1500, 2, 0.7, 0.0), (4000, 18, 0.5, 0.0), (64, 1, 0.0, 0.0), (200, 4, 0.5, 2.0),
                                                   (9000, 18, 0.5, 0.0)])        # above one workgroup's LDS limit: launches cut at class boundaries
def test_multiclass_nms_matches_oracle(n, n_cls, thr, score_thr, fast):
    from unidet3d_amd import ops
    rng = np.random.default_rng(n * 31 + n_cls)
    boxes = _boxes(rng, n)
    if n > 10:
        boxes[5] = boxes[4]                     # exact duplicates and a zero-area box
        boxes[7, 3:5] = 0
    scores = np.sort(rng.uniform(0.01, 1.0, n).astype(F32))[::-1].copy()
    labels = rng.integers(0, n_cls, n)
    ob, os_, ol = pp.multiclass_nms(boxes, scores, labels, thr, score_thr, fast)
    gb, gs, gl = ops.nms_multiclass(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV),
                                    torch.from_numpy(labels).to(DEV), thr, score_thr, fast)
    assert gl.cpu().numpy().tolist() == ol.tolist()
    assert np.array_equal(gs.cpu().numpy(), os_) and np.array_equal(gb.cpu().numpy(), ob)


@pytest.mark.parametrize('n,n_cls,thr', [(1, 1, 0.5), (400, 6, 0.5), (1000, 18, 0.3), (1300, 2, 0.6), (3000, 18, 0.5), (8000, 18, 0.5)])
def test_rotated_nms_matches_oracle(n, n_cls, thr):
    """mmcv nms3d path (7-dof boxes).  The kernel sums the clipped edges in fp32, the oracle intersects polygons in fp64: a pair
    whose IoU lies within 1e-4 of the threshold may legitimately flip, so such inputs are nudged away from it first."""
    from unidet3d_amd import ops
    from oracle import rotated_iou as ri
    rng = np.random.default_rng(n + n_cls)
    boxes = np.concatenate([_boxes(rng, n, spread=3.0), rng.uniform(-3.2, 3.2, (n, 1)).astype(F32)], 1)
    if n > 10:
        boxes[5] = boxes[4]                     # identical boxes
        boxes[7, :2] = boxes[6, :2]; boxes[7, 6] = boxes[6, 6] + np.pi / 2      # same centre, perpendicular
    scores = np.sort(rng.uniform(0.01, 1.0, n).astype(F32))[::-1].copy()
    labels = rng.integers(0, n_cls, n)
    b5 = torch.from_numpy(boxes[:, [0, 1, 3, 4, 6]]).double()
    cor = ri.box2corners(b5)
    for c in np.unique(labels):                 # drop one box of every pair that sits on the threshold
        idx = np.nonzero(labels == c)[0]
        if len(idx) < 2:
            continue
        ii, jj = np.triu_indices(len(idx), 1)
        inter = ri.oriented_box_intersection_2d(cor[idx[ii]], cor[idx[jj]]).numpy()
        ar = (b5[:, 2] * b5[:, 3]).numpy()
        iou = inter / np.maximum(ar[idx[ii]] + ar[idx[jj]] - inter, 1e-8)
        for j in idx[jj][np.abs(iou - thr) < 1e-4]:
            boxes[j, :2] += 40.0
    ob, os_, ol = pp.multiclass_nms(boxes, scores, labels, thr, 0.0)
    gb, gs, gl = ops.nms_multiclass(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), torch.from_numpy(labels).to(DEV), thr, 0.0)
    assert gl.cpu().numpy().tolist() == ol.tolist()
    assert np.array_equal(gs.cpu().numpy(), os_) and np.array_equal(gb.cpu().numpy(), ob)


def test_nms_rejects_too_many_boxes_and_bad_shapes():
    from unidet3d_amd import _lib as L, ops
    b = torch.zeros(5000, 6, device=DEV); s = torch.linspace(1, 0.1, 5000, device=DEV); l = torch.zeros(5000, dtype=torch.long, device=DEV)
    with pytest.raises(L.U3DError):
        ops.nms_bev_multiclass(b, s, l, 0.5, 0.0)
    with pytest.raises(ValueError):
        ops.nms_multiclass(torch.zeros(4, 5, device=DEV), s[:4], l[:4], 0.5, 0.0)


@pytest.mark.parametrize('n_pts,n_sp,n_box', [(5000, 60, 1), (20000, 300, 130), (100000, 1500, 700), (1000, 1000, 65), (10, 3, 0)])
def test_trim_boxes_matches_oracle(n_pts, n_sp, n_box):
    from unidet3d_amd import ops
    rng = np.random.default_rng(n_pts + n_box)
    pts = rng.uniform(-3, 3, (n_pts, 6)).astype(F32)
    # superpoints = spatial cells (compact clusters, like the segmentator's output), ragged sizes, some empty ids
    sp = ((pts[:, 0] > 0) * 1 + (pts[:, 1] > 0) * 2 + rng.integers(0, max(n_sp // 4, 1), n_pts) * 4) % n_sp
    boxes = _boxes(rng, n_box, spread=2.5)
    if n_box > 2:
        boxes[1, :3] = 50.0                      # a box without any point
        boxes[2, 3:] = 20.0                      # a box holding every point
    want = pp.trim_boxes(pts, sp, boxes, 0.18, 0.81)
    tp, tsp = torch.from_numpy(pts).to(DEV), torch.from_numpy(sp).to(DEV)
    off, lst = ops.csr_build(tsp, n_sp)
    got = ops.trim_boxes_by_superpoints(tp, off, lst, n_sp, torch.from_numpy(boxes).to(DEV), 0.18, 0.81).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def test_predict_end_to_end_matches_oracle_postprocessing():
    """UniDet3D.predict on a synthetic scene: its boxes / labels / scores equal the oracle's NMS + trimming applied to the
    same decoder outputs (the decoder itself is covered against the oracle model in test_gpu_model.py)."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import build_model, scannet_model_cfg
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = scannet_model_cfg()
    cfg['decoder']['num_layers'] = 2
    model = build_model(cfg)
    fill_state_dict(model, tag0=3000, scale=0.06)
    model = model.to(DEV).eval()
    model.voxel_size = 0.05
    sc = make_scene(11, n_points=20_000)
    inputs, samples = make_batch_inputs([sc], DEV)
    seen, orig = {}, model.predict_by_feat
    model.predict_by_feat = lambda out, *a, **k: (seen.update(out=out), orig(out, *a, **k))[1]      # the decoder output predict() used
    with torch.no_grad():
        res = model.predict(inputs, samples)[0].pred_instances_3d
    out = seen['out']
    cls_preds, bboxes = out['cls_preds'][0], out['bboxes'][0]
    scores = torch.softmax(cls_preds, -1)[:, :-1]
    nc = scores.shape[1]
    s, idx = scores.flatten().topk(min(1000, scores.numel()), sorted=True)
    lab, q = (idx % nc).cpu().numpy(), torch.div(idx, nc, rounding_mode='floor')
    nb, ns, nl = pp.multiclass_nms(bboxes[q].cpu().numpy(), s.cpu().numpy(), lab, 0.5, 0.0)
    want = pp.trim_boxes(sc.points[:, :3], sc.superpoints, nb, 0.18, 0.81)
    assert len(nl) > 0
    assert res.labels_3d.cpu().numpy().tolist() == nl.tolist()
    assert np.array_equal(res.scores_3d.cpu().numpy(), ns)
    want[:, 2] += want[:, 5] * F32(-0.5)        # DepthInstance3DBoxes(origin=(0.5, 0.5, 0.5)) stores the bottom centre (mmdet3d convention)
    assert np.array_equal(res.bboxes_3d.tensor.cpu().numpy(), want, equal_nan=True)

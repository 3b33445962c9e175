"""Oracle of the inference post-processing (oracle/postproc.py) on hand-computable cases (CPU)."""
import numpy as np

from oracle import postproc as pp

F32 = np.float32


def _box(x, y, z, dx, dy, dz):
    return np.array([x, y, z, dx, dy, dz], F32)


def test_iou_normal_known_values():
    a = _box(0, 0, 0, 1, 1, 1)
    assert pp.iou_normal(a, a) == F32(1)
    assert pp.iou_normal(a, _box(0.5, 0, 5, 1, 1, 9)) == F32(F32(0.5) / F32(1.5))     # z / dz ignored
    assert pp.iou_normal(a, _box(2, 2, 0, 1, 1, 1)) == F32(0)
    assert pp.iou_normal(_box(0, 0, 0, 0, 0, 0), _box(0, 0, 0, 0, 0, 0)) == F32(0)      # eps guard


def test_nms_greedy_order_and_threshold():
    boxes = np.stack([_box(0, 0, 0, 1, 1, 1), _box(0.1, 0, 0, 1, 1, 1), _box(0.6, 0, 0, 1, 1, 1), _box(5, 5, 0, 1, 1, 1)])
    scores = np.array([0.5, 0.9, 0.8, 0.1], F32)
    # visit order 1, 2, 0, 3: box 2 overlaps box 1 with IoU 0.5/1.5 = 0.33 (kept at thr 0.5), box 0 has IoU 0.9/1.1 = 0.82 (dropped)
    assert pp.nms3d_normal(boxes, scores, 0.5).tolist() == [1, 2, 3]
    assert pp.nms3d_normal(boxes, scores, 0.3).tolist() == [1, 3]


def test_multiclass_nms_output_order():
    boxes = np.stack([_box(0, 0, 0, 1, 1, 1)] * 4)
    scores = np.array([0.9, 0.8, 0.7, 0.6], F32)
    labels = np.array([3, 1, 3, 1])
    b, s, l = pp.multiclass_nms(boxes, scores, labels, 0.5, 0.0)
    assert l.tolist() == [1, 3] and s.tolist() == [F32(0.8), F32(0.9)]          # classes ascending, identical boxes suppressed
    b, s, l = pp.multiclass_nms(boxes, scores, labels, 0.5, 0.85)
    assert l.tolist() == [3]                                                    # score threshold is applied per class first
    b, s, l = pp.multiclass_nms(boxes, scores, labels, 0.5, 1.0)
    assert b.shape == (0, 6) and len(s) == 0


def test_trim_delete_add_and_empty():
    # superpoint 0: 4 points, 3 inside the box -> ratio 0.75: kept as is (only its inside points count)
    # superpoint 1: 10 points, 9 inside -> ratio 0.9 > up: the outside point is added
    # superpoint 2: 10 points, 1 inside -> ratio 0.1 < low: its inside point is deleted
    pts, sp = [], []
    for i in range(3):
        pts.append([0.1 * i, 0.0, 0.0]); sp.append(0)
    pts.append([3.0, 0.0, 0.0]); sp.append(0)
    for i in range(9):
        pts.append([0.0, 0.05 * i, 0.0]); sp.append(1)
    pts.append([0.0, 2.0, 0.0]); sp.append(1)
    pts.append([0.0, 0.0, 0.4]); sp.append(2)
    for i in range(9):
        pts.append([0.0, 0.0, 5.0 + i]); sp.append(2)
    pts, sp = np.array(pts, F32), np.array(sp)
    box = _box(0, 0, 0, 1, 1, 1)[None]
    out = pp.trim_boxes(pts, sp, box, 0.18, 0.81)
    mn = np.array([0.0, 0.0, 0.0], F32); mx = np.array([0.2, 2.0, 0.0], F32)
    assert np.array_equal(out[0, :3], (mx + mn) / F32(2)) and np.array_equal(out[0, 3:], mx - mn)
    far = _box(100, 100, 100, 1, 1, 1)[None]
    out = pp.trim_boxes(pts, sp, far, 0.18, 0.81)
    assert np.isnan(out[0, :3]).all() and np.isneginf(out[0, 3:]).all()         # the reference's behaviour for an empty box


def test_inside_is_strict_and_uses_the_reference_rounding():
    box = _box(0.1, 0, 0, 1, 1, 1)[None]
    pts = np.array([[0.6, 0, 0], [0.59999, 0, 0], [-0.4, 0, 0]], F32)
    ins = pp.inside_boxes(pts, box)[0]
    assert ins.tolist() == [False, True, False]                                 # faces are outside (distance must be > 0)


def test_topk_instances_labels():
    sc = np.array([[0.1, 0.7], [0.6, 0.2], [0.3, 0.9]], F32)
    s, l, q = pp.topk_instances(sc, 4)
    assert s.tolist() == [F32(0.9), F32(0.7), F32(0.6), F32(0.3)] and l.tolist() == [1, 1, 0, 0] and q.tolist() == [2, 0, 1, 2]


def test_aligned_3d_nms_uses_volume_iou_and_drops_degenerate_duplicates():
    a = pp.bbox_to_loss(np.stack([_box(0, 0, 0, 1, 1, 1), _box(0.5, 0, 0, 1, 1, 1), _box(0, 0, 0.9, 1, 1, 1), _box(3, 3, 3, 0, 0, 0), _box(3, 3, 3, 0, 0, 0)]))
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5], F32)
    cls = np.zeros(5, np.int64)
    # box 1: IoU 0.5/1.5 = 0.33 with box 0; box 2 overlaps box 0 only by 0.1 in z (IoU 0.1/1.9); boxes 3/4 have zero volume:
    # 3 survives (0/vol = 0 against the others), 4 meets 3 with 0/0 = NaN -> dropped
    assert pp.aligned_3d_nms(a, scores, cls, 0.25).tolist() == [0, 2, 3]
    assert pp.aligned_3d_nms(a, scores, cls, 0.5).tolist() == [0, 1, 2, 3]
    b, s, l = pp.multiclass_nms(np.stack([_box(0, 0, 0, 1, 1, 1), _box(0, 0, 0.8, 1, 1, 1)]), np.array([0.9, 0.8], F32), np.array([2, 2]), 0.5, 0.0, fast_nms=False)
    assert len(l) == 2                      # BEV NMS would have merged them (identical footprint), the 3-D IoU is 0.2/1.8
    b, s, l = pp.multiclass_nms(np.stack([_box(0, 0, 0, 1, 1, 1), _box(0, 0, 0.8, 1, 1, 1)]), np.array([0.9, 0.8], F32), np.array([2, 2]), 0.5, 0.0, fast_nms=True)
    assert len(l) == 1


def test_rotated_nms_uses_the_rotated_footprints():
    b7 = lambda x, y, dx, dy, a: np.array([x, y, 0.0, dx, dy, 1.0, a], F32)
    long_a, long_b = b7(0, 0, 4, 1, 0.0), b7(0, 0, 4, 1, np.pi / 2)            # a cross: intersection 1, IoU 1/7
    boxes = np.stack([long_a, long_b, b7(0.1, 0, 4, 1, 0.02), b7(9, 9, 1, 1, 0.3)])
    scores = np.array([0.9, 0.8, 0.7, 0.6], F32)
    assert pp.nms3d_rotated(boxes, scores, 0.5).tolist() == [0, 1, 3]           # the nearly parallel box 2 is suppressed by box 0
    assert pp.nms3d_rotated(boxes, scores, 0.1).tolist() == [0, 3]              # at 0.1 the crossing box (IoU 0.143) goes too
    b, s, l = pp.multiclass_nms(boxes, scores, np.array([1, 1, 2, 2]), 0.5, 0.0)
    assert l.tolist() == [1, 1, 2, 2] and b.shape == (4, 7)                     # different classes never suppress each other

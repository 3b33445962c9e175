"""Indoor detection evaluation (mAP / mAR at IoU thresholds) -- SURVEY.md section 8f rank 4.

Same entry point, argument meaning and result dictionary as the reference's ``indoor_eval``
(unidet3d/indoor_eval.py:203-302; ``eval_det_cls`` :56-161, ``eval_map_recall`` :164-200,
``average_precision`` :8-53) so that ``IndoorMetric_.compute_metrics`` (unidet3d/indoor_metric.py:65-100)
can call it unchanged.  Differences below the surface: the IoU of every (detection, ground truth) pair of an
image is ONE batched tensor op on whatever device the boxes live on (the reference builds a box object per
detection and calls mmcv's rotated-IoU kernel per image and class), and the true/false-positive sweep works on
precomputed best-match arrays instead of nested Python loops.  Boxes: ``DepthInstance3DBoxes`` (bottom-centre
``tensor`` [n, 6 or 7]).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch


def average_precision(recalls: np.ndarray, precisions: np.ndarray, mode: str = 'area') -> np.ndarray:
    """VOC-style AP of one or several (recall, precision) curves; 'area' (all points) or '11points'."""
    recalls, precisions = np.atleast_2d(recalls), np.atleast_2d(precisions)
    assert recalls.shape == precisions.shape
    n = recalls.shape[0]
    ap = np.zeros(n, dtype=np.float32)
    if mode == 'area':
        mrec = np.concatenate((np.zeros((n, 1), recalls.dtype), recalls, np.ones((n, 1), recalls.dtype)), 1)
        mpre = np.concatenate((np.zeros((n, 1), recalls.dtype), precisions, np.zeros((n, 1), recalls.dtype)), 1)
        mpre = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]       # precision envelope (monotone from the right)
        for i in range(n):
            ind = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, ind + 1] - mrec[i, ind]) * mpre[i, ind + 1])
    elif mode == '11points':
        for i in range(n):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i, :] >= thr]
                ap[i] += precs.max() if precs.size > 0 else 0
            ap /= 11          # inside the scale loop, as the reference has it (:47-48)
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap


def boxes_iou_3d(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """[n1, n2] 3-D IoU of depth boxes in bottom-centre form (x, y, z_bottom, dx, dy, dz[, yaw]) -- what
    ``BaseInstance3DBoxes.overlaps`` returns: BEV intersection x height overlap over the union of the volumes."""
    n1, n2 = t1.shape[0], t2.shape[0]
    if n1 == 0 or n2 == 0:
        return t1.new_zeros((n1, n2))
    a, b = t1[:, None].float(), t2[None].float()
    zlo = torch.max(a[..., 2], b[..., 2])
    zhi = torch.min(a[..., 2] + a[..., 5], b[..., 2] + b[..., 5])
    h = (zhi - zlo).clamp(min=0)
    yaw = (t1.shape[1] == 7 and bool((t1[:, 6] != 0).any())) or (t2.shape[1] == 7 and bool((t2[:, 6] != 0).any()))
    if yaw:
        from .criterion import _box2corners, _oriented_box_intersection_2d
        z1 = t1.new_zeros(n1, 1) if t1.shape[1] == 6 else t1[:, 6:7]
        z2 = t2.new_zeros(n2, 1) if t2.shape[1] == 6 else t2[:, 6:7]
        r1 = torch.cat((t1[:, [0, 1, 3, 4]], z1), 1)[:, None].expand(n1, n2, 5)
        r2 = torch.cat((t2[:, [0, 1, 3, 4]], z2), 1)[None].expand(n1, n2, 5)
        bev = _oriented_box_intersection_2d(_box2corners(r1.float()), _box2corners(r2.float()))
    else:
        wx = (torch.min(a[..., 0] + a[..., 3] / 2, b[..., 0] + b[..., 3] / 2) - torch.max(a[..., 0] - a[..., 3] / 2, b[..., 0] - b[..., 3] / 2)).clamp(min=0)
        wy = (torch.min(a[..., 1] + a[..., 4] / 2, b[..., 1] + b[..., 4] / 2) - torch.max(a[..., 1] - a[..., 4] / 2, b[..., 1] - b[..., 4] / 2)).clamp(min=0)
        bev = wx * wy
    inter = bev * h
    v1 = (a[..., 3] * a[..., 4] * a[..., 5])
    v2 = (b[..., 3] * b[..., 4] * b[..., 5])
    return inter / (v1 + v2 - inter).clamp(min=1e-8)


def _box_tensor(b) -> torch.Tensor:
    t = b.tensor if hasattr(b, 'tensor') else torch.as_tensor(b)
    return t.reshape(-1, t.shape[-1]) if t.numel() else t.reshape(0, 7)


def eval_det_cls(det_img: np.ndarray, det_score: np.ndarray, iou_max: np.ndarray, jmax: np.ndarray, n_gt_img: Dict[int, int],
                 iou_thr: Sequence[float]):
    """Precision / recall / AP of one class.  det_* are per detection (image id, confidence, best IoU with a GT of this class
    in its image, index of that GT); n_gt_img: image id -> number of GTs of the class."""
    npos = sum(n_gt_img.values())
    order = np.argsort(-det_score)
    det_img, iou_max, jmax = det_img[order], iou_max[order], jmax[order]
    nd = len(order)
    ret = []
    for thr in iou_thr:
        taken = {img: np.zeros(n, bool) for img, n in n_gt_img.items()}
        tp, fp = np.zeros(nd), np.zeros(nd)
        for d in range(nd):
            if iou_max[d] > thr and not taken[det_img[d]][jmax[d]]:
                tp[d] = 1.
                taken[det_img[d]][jmax[d]] = True
            else:
                fp[d] = 1.
        fp, tp = np.cumsum(fp), np.cumsum(tp)
        with np.errstate(divide='ignore', invalid='ignore'):
            recall = tp / float(npos)
        precision = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        ret.append((recall, precision, average_precision(recall, precision)))
    return ret


def indoor_eval(gt_annos: List[dict], dt_annos: List[dict], metric: Sequence[float], label2cat, logger=None, box_mode_3d=None):
    """gt_annos[i]: ``gt_bboxes_3d`` (boxes), ``gt_labels_3d`` (sequence of int); dt_annos[i]: ``labels_3d``, ``bboxes_3d``,
    ``scores_3d``.  Returns ``{<cat>_AP_<thr>, mAP_<thr>, <cat>_rec_<thr>, mAR_<thr>}`` like the reference."""
    assert len(dt_annos) == len(gt_annos)
    classes: List[int] = []          # insertion order of the reference's ``gt`` dict: per image, detections first, then GTs
    has_pred = set()
    dets = {}                        # class -> lists
    gts = {}                         # class -> {img: count}
    for img, (g, d) in enumerate(zip(gt_annos, dt_annos)):
        dl = torch.as_tensor(d['labels_3d']).cpu().numpy().astype(np.int64)
        ds = torch.as_tensor(d['scores_3d']).cpu().numpy()
        gl = np.asarray([int(x) for x in g['gt_labels_3d']], dtype=np.int64)
        db, gb = _box_tensor(d['bboxes_3d']), _box_tensor(g['gt_bboxes_3d'])
        iou = boxes_iou_3d(db, gb.to(db.device)).cpu().numpy() if len(dl) and len(gl) else np.zeros((len(dl), len(gl)), np.float32)
        for c in dl.tolist():
            if c not in gts:
                gts[c] = {}
                classes.append(c)
            has_pred.add(c)
            gts[c].setdefault(img, 0)
        for c in gl.tolist():
            if c not in gts:
                gts[c] = {}
                classes.append(c)
            gts[c][img] = gts[c].get(img, 0) + 1
        for c in np.unique(dl).tolist():
            sel = dl == c
            gsel = np.nonzero(gl == c)[0]
            sub = iou[sel][:, gsel]
            rec = dets.setdefault(c, dict(img=[], score=[], iou=[], j=[]))
            rec['img'].append(np.full(int(sel.sum()), img)); rec['score'].append(ds[sel])
            if len(gsel):
                rec['iou'].append(sub.max(1)); rec['j'].append(sub.argmax(1))        # first maximum, like the strict '>' scan (:137-139)
            else:
                rec['iou'].append(np.full(int(sel.sum()), -np.inf)); rec['j'].append(np.zeros(int(sel.sum()), np.int64))
    rec_c, prec_c, ap_c = {}, {}, {}
    for c in classes:
        if c in has_pred:
            r = dets[c]
            res = eval_det_cls(np.concatenate(r['img']), np.concatenate(r['score']), np.concatenate(r['iou']), np.concatenate(r['j']),
                               gts[c], metric)
        else:
            res = [(np.zeros(1), np.zeros(1), np.zeros(1))] * len(metric)
        rec_c[c], prec_c[c], ap_c[c] = zip(*res)
    ret = {}
    rows = [['classes'] + [label2cat[c] for c in classes] + ['Overall']]
    for i, thr in enumerate(metric):
        aps = [ap_c[c][i] for c in classes]
        for c, a in zip(classes, aps):
            ret[f'{label2cat[c]}_AP_{thr:.2f}'] = float(a[0])
        ret[f'mAP_{thr:.2f}'] = float(np.nanmean(aps))
        recs = [rec_c[c][i][-1] for c in classes]
        for c, r in zip(classes, recs):
            ret[f'{label2cat[c]}_rec_{thr:.2f}'] = float(r)
        ret[f'mAR_{thr:.2f}'] = float(np.nanmean(recs))
        rows.append([f'AP_{thr:.2f}'] + [f'{float(a[0]):.4f}' for a in aps] + [f"{ret[f'mAP_{thr:.2f}']:.4f}"])
        rows.append([f'AR_{thr:.2f}'] + [f'{float(r):.4f}' for r in recs] + [f"{ret[f'mAR_{thr:.2f}']:.4f}"])
    if logger is not None:
        table = '\n'.join('  '.join(f'{cell:>14}' for cell in col) for col in zip(*rows))
        (logger.info if hasattr(logger, 'info') else print)('\n' + table)
    return ret


class IndoorMetric:
    """Accumulates (annotation, prediction) pairs per dataset and evaluates each dataset with ``indoor_eval`` -- the part of
    ``IndoorMetric_`` (unidet3d/indoor_metric.py:14-100) that carries arithmetic; the mmengine ``BaseMetric`` plumbing is not here."""

    def __init__(self, datasets: List[str], datasets_classes: List[List[str]], iou_thr=(0.25, 0.5)):
        self.datasets, self.datasets_classes, self.iou_thr = datasets, datasets_classes, list(iou_thr)
        self.results: List = []

    def process(self, eval_ann_info: dict, pred: dict):
        self.results.append((eval_ann_info, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in pred.items()}))

    def compute_metrics(self) -> Dict[str, dict]:
        out = {}
        for i, name in enumerate(self.datasets):
            anns = [a for a, p in self.results if p['dataset'] == name]
            preds = [p for a, p in self.results if p['dataset'] == name]
            out[name] = indoor_eval(anns, preds, self.iou_thr, self.datasets_classes[i])
        return out

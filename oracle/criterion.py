"""CPU ORACLE (test infrastructure, NOT product code) -- loss driver restated.

Follows unidet3d/criterion.py:44-320 (UniDet3DCriterion, UniMatcher, the two
cost classes, _bbox_to_loss) and unidet3d/axis_aligned_iou_loss.py:14-53.
External arithmetic restated from its published form (parity unpinned by the
reference, which holds no tests):
  * mmdet3d 1.4.0 ``AxisAlignedBboxOverlaps3D`` (is_aligned=True, eps=1e-6)
  * mmdet 3.3.0 ``weighted_loss`` with reduction='none', weight=None (identity)
  * mmdet3d ``DepthInstance3DBoxes(origin=(.5,.5,.5))``: stores the bottom
    centre, ``gravity_center`` adds h/2 back.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class OBoxes:
    """DepthInstance3DBoxes stand-in: [n, 6] axis-aligned or [n, 7] with a heading (``with_yaw``)."""

    def __init__(self, centers_sizes: torch.Tensor, with_yaw: bool = False):
        t = centers_sizes.clone()
        if t.numel():
            t[:, 2] = t[:, 2] - t[:, 5] * 0.5       # origin (.5,.5,.5) -> (.5,.5,0)
        self.tensor = t
        self.with_yaw = with_yaw

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = c[:, 2] + self.tensor[:, 5] * 0.5
        return c

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, idx):
        b = OBoxes.__new__(OBoxes)
        b.tensor = self.tensor[idx]
        b.with_yaw = self.with_yaw
        return b


class OInst:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __len__(self):
        return len(self.labels_3d)


def aligned_iou_3d(b1, b2, eps=1e-6):
    a1 = (b1[..., 3] - b1[..., 0]) * (b1[..., 4] - b1[..., 1]) * (b1[..., 5] - b1[..., 2])
    a2 = (b2[..., 3] - b2[..., 0]) * (b2[..., 4] - b2[..., 1]) * (b2[..., 5] - b2[..., 2])
    lt = torch.max(b1[..., :3], b2[..., :3])
    rb = torch.min(b1[..., 3:], b2[..., 3:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
    union = torch.max(a1 + a2 - overlap, overlap.new_tensor([eps]))
    return overlap / union


def axis_aligned_diou_loss(pred, target):      # axis_aligned_iou_loss.py:14-53
    iou_loss = 1 - aligned_iou_3d(pred, target)
    xp1, yp1, zp1, xp2, yp2, zp2 = pred.split(1, dim=-1)
    xt1, yt1, zt1, xt2, yt2, zt2 = target.split(1, dim=-1)
    r2 = ((xp1 + xp2) / 2 - (xt1 + xt2) / 2) ** 2 + ((yp1 + yp2) / 2 - (yt1 + yt2) / 2) ** 2 + \
         ((zp1 + zp2) / 2 - (zt1 + zt2) / 2) ** 2
    c2 = (torch.minimum(xp1, xt1) - torch.maximum(xp2, xt2)) ** 2 + \
         (torch.minimum(yp1, yt1) - torch.maximum(yp2, yt2)) ** 2 + \
         (torch.minimum(zp1, zt1) - torch.maximum(zp2, zt2)) ** 2
    # NOTE ``[:, 0]`` is the reference's own indexing (:51): for the [n, n_gt, 6]
    # cost-matrix call it broadcasts gt 0's centre term over all gts.
    return iou_loss + (r2 / c2)[:, 0]


def bbox_to_loss(bbox):                        # criterion.py:180-198
    if bbox.shape[-1] != 6:
        return bbox
    return torch.stack((bbox[..., 0] - bbox[..., 3] / 2, bbox[..., 1] - bbox[..., 4] / 2,
                        bbox[..., 2] - bbox[..., 5] / 2, bbox[..., 0] + bbox[..., 3] / 2,
                        bbox[..., 1] + bbox[..., 4] / 2, bbox[..., 2] + bbox[..., 5] / 2), dim=-1)


def box_cost_loss(pred, target):
    """bbox loss / cost on _bbox_to_loss boxes: axis-aligned DIoU (6 columns) or the rotated DIoU of
    unidet3d/rotated_iou_loss.py:14-82 (7 columns; [..., 7] inputs are flattened to the [N, 7] form it takes)."""
    if pred.shape[-1] == 7:
        from . import rotated_iou as ri
        return ri.rotated_diou_3d_loss(pred.reshape(-1, 7), target.reshape(-1, 7)).reshape(pred.shape[:-1])
    return axis_aligned_diou_loss(pred, target)


@torch.no_grad()
def uni_matcher(scores, bboxes, gt_labels, gt_bboxes, query_masks, topk,
                w_cls=0.5, w_box=2.0, inf=1e8):   # criterion.py:272-320
    n_gts = len(gt_labels)
    if n_gts == 0:
        return gt_labels.new_empty((0,)), gt_labels.new_empty((0,))
    c_cls = -scores.softmax(-1)[:, gt_labels] * w_cls                       # :222-224
    pb = bboxes.unsqueeze(1).repeat(1, n_gts, 1)
    gb = gt_bboxes.unsqueeze(0).repeat(bboxes.shape[0], 1, 1)
    c_box = box_cost_loss(bbox_to_loss(pb), bbox_to_loss(gb)) * w_box      # :256-270
    cost = torch.stack([c_cls, c_box]).sum(0)
    cost = torch.where(query_masks.T, cost, torch.tensor(inf, dtype=cost.dtype))
    values = torch.topk(cost, topk + 1, dim=0, sorted=True, largest=False).values[-1:, :]
    ids = torch.argwhere(cost < values)
    return ids[:, 0], ids[:, 1]


def _gt_box_rows(b):
    """criterion.py:87-91 / :117-122: gravity centre + (size[, heading])."""
    return torch.cat((b.gravity_center, b.tensor[:, 3:] if b.with_yaw else b.tensor[:, 3:6]), dim=1)


def layer_loss(cls_preds, pred_bboxes, insts, topk=6, loss_weight=(0.5, 1.0),
               non_object_weight=0.1, dataset_weight=1.0):   # criterion.py:44-143
    """``topk`` / ``dataset_weight``: one value, or one per scene (the reference looks them up by dataset name :82,:104-105)."""
    n = len(insts)
    topk = list(topk) if isinstance(topk, (list, tuple)) else [topk] * n
    dw = list(dataset_weight) if isinstance(dataset_weight, (list, tuple)) else [dataset_weight] * n
    indices = []
    for i, inst in enumerate(insts):
        indices.append(uni_matcher(cls_preds[i], pred_bboxes[i], inst.labels_3d, _gt_box_rows(inst.bboxes_3d),
                                   inst.query_masks, topk[i]))
    cls_losses = []
    for w, cls_pred, inst, (iq, ig) in zip(dw, cls_preds, insts, indices):
        nc = cls_pred.shape[1] - 1
        tgt = cls_pred.new_full((len(cls_pred),), nc, dtype=torch.long)
        tgt[iq] = inst.labels_3d[ig]
        cls_losses.append(w * F.cross_entropy(
            cls_pred, tgt, cls_pred.new_tensor([1] * nc + [non_object_weight])))
    cls_loss = torch.mean(torch.stack(cls_losses))
    box_losses = []
    for w, bbox, inst, (iq, ig) in zip(dw, pred_bboxes, insts, indices):
        if len(inst) == 0 or len(iq) == 0:
            continue
        tb = _gt_box_rows(inst.bboxes_3d[ig])
        box_losses.append(w * box_cost_loss(bbox_to_loss(bbox[iq]), bbox_to_loss(tb)).mean())
    box_loss = torch.stack(box_losses).mean() if box_losses else 0
    return loss_weight[0] * cls_loss + loss_weight[1] * box_loss


def criterion(pred, insts, **kw):              # criterion.py:145-178 (iter_matcher=True)
    loss = layer_loss(pred['cls_preds'], pred['bboxes'], insts, **kw)
    for aux in pred['aux_outputs']:
        loss = loss + layer_loss(aux['cls_preds'], aux['bboxes'], insts, **kw)
    return loss


def get_targets(points, gt_centers, topk):     # unidet3d.py:371-409
    """[n_boxes, n_points] bool: every point goes to its nearest box centre among the boxes that count it among their
    ``topk`` nearest points (strictly closer than the (topk+1)-th)."""
    float_max = points.new_tensor(1e8)
    n_boxes = len(gt_centers)
    d = torch.sum(torch.pow(gt_centers[None].expand(len(points), n_boxes, 3) - points[:, None].expand(len(points), n_boxes, 3), 2), dim=-1)
    kth = torch.topk(d, min(topk + 1, len(d)), largest=False, dim=0).values[-1]
    d = torch.where(d < kth.unsqueeze(0), d, float_max)
    min_values, min_ids = d.min(dim=1)
    min_inds = torch.where(min_values < float_max, min_ids, n_boxes)
    return F.one_hot(min_inds, num_classes=n_boxes + 1)[:, :-1].bool().T


# ---- GT preparation (unidet3d.py:220-275, transforms_3d.py:197-215) -------------
def gt_from_scene(points_xyz_shifted, instance_mask, labels, superpoints):
    """points already shifted by -min (unidet3d.py:301-302). Returns OInst with
    labels_3d, bboxes_3d (from masks, :220-256), sp_masks [n_inst, S]."""
    n_inst = len(labels)
    boxes, sp_masks = [], []
    S = int(superpoints.max()) + 1
    cnt = torch.bincount(superpoints, minlength=S).clamp(min=1).float()
    keep = []
    for j in range(n_inst):
        m = instance_mask == j
        if m.sum() == 0:
            continue
        keep.append(j)
        op = points_xyz_shifted[m]
        lo, hi = op.min(0).values, op.max(0).values
        boxes.append(torch.cat(((hi + lo) / 2, hi - lo)))
        frac = torch.zeros(S).index_add_(0, superpoints, m.float()) / cnt
        sp_masks.append(frac > 0.5)
    boxes = torch.stack(boxes) if boxes else torch.zeros(0, 6)
    sp_masks = torch.stack(sp_masks) if sp_masks else torch.zeros(0, S, dtype=torch.bool)
    return OInst(labels_3d=labels[keep], bboxes_3d=OBoxes(boxes), sp_masks=sp_masks,
                 query_masks=sp_masks)

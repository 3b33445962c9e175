"""Light containers standing in for the mmengine / mmdet3d structures the hot path touches
(``InstanceData_`` unidet3d/structures.py:5-25, ``DepthInstance3DBoxes`` as used at
unidet3d/unidet3d.py:249-255,326-330 and unidet3d/criterion.py:87-91,117-122, and the
``Det3DDataSample`` fields read by ``UniDet3D.loss`` unidet3d/unidet3d.py:307-347)."""
from __future__ import annotations

import torch


class InstanceData_:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __len__(self):
        for v in self.__dict__.values():
            if hasattr(v, '__len__'):
                return len(v)
        return 0

    def keys(self):
        return list(self.__dict__.keys())


class DepthInstance3DBoxes:
    """Axis-aligned (box_dim=6) or yawed (box_dim=7) boxes. Constructed from gravity centres
    (origin=(0.5,0.5,0.5)); stores the bottom centre like mmdet3d does, so ``gravity_center``
    reproduces its z - h/2 + h/2 round trip."""

    def __init__(self, tensor, with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5)):
        t = torch.as_tensor(tensor, dtype=torch.float32).reshape(-1, box_dim).clone()
        if origin != (0.5, 0.5, 0):
            dst = t.new_tensor((0.5, 0.5, 0))
            src = t.new_tensor(origin)
            t[:, :3] += t[:, 3:6] * (dst - src)
        self.tensor = t
        self.with_yaw = with_yaw
        self.box_dim = box_dim
        self.gt_rows = None

    def cache_gt_rows(self):
        """(gravity centre, size[, heading]) rows -- what the criterion packs per scene (criterion._gt_boxes) -- computed once for
        all boxes; row slices of this object (``__getitem__`` with a slice) carry the matching row views."""
        self.gt_rows = None
        self.gt_rows = torch.cat((self.gravity_center, self.tensor[:, 3:] if self.with_yaw else self.tensor[:, 3:6]), dim=1)
        return self.gt_rows

    @property
    def gravity_center(self):
        if self.gt_rows is not None and self.gt_rows.shape[0] == self.tensor.shape[0]:
            return self.gt_rows[:, :3]                     # computed once (cache_gt_rows): the same values
        # (x, y, z_bottom + h / 2): three launches (mul, add, cat) instead of the five of zeros_like + two slice assignments
        return torch.cat((self.tensor[:, :2], self.tensor[:, 2:3] + self.tensor[:, 5:6] * 0.5), dim=1)

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, idx):
        b = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
        b.tensor = self.tensor[idx].reshape(-1, self.box_dim)
        b.with_yaw, b.box_dim = self.with_yaw, self.box_dim
        b.gt_rows = self.gt_rows[idx] if self.gt_rows is not None and isinstance(idx, slice) else None
        return b

    def to(self, device):
        b = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
        b.tensor = self.tensor.to(device)
        b.with_yaw, b.box_dim = self.with_yaw, self.box_dim
        b.gt_rows = None
        return b


class PointSegData:
    """gt_pts_seg: pts_instance_mask int64 [N], sp_pts_mask int64 [N]."""

    def __init__(self, pts_instance_mask=None, sp_pts_mask=None, pts_semantic_mask=None):
        self.pts_instance_mask = pts_instance_mask
        self.sp_pts_mask = sp_pts_mask
        self.pts_semantic_mask = pts_semantic_mask


class Det3DDataSample:
    def __init__(self, lidar_path, gt_pts_seg: PointSegData, gt_instances_3d: InstanceData_ = None):
        self.lidar_path = lidar_path
        self.gt_pts_seg = gt_pts_seg
        self.gt_instances_3d = gt_instances_3d if gt_instances_3d is not None else InstanceData_()
        self.pred_instances_3d = None

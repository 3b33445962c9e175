"""Host-side batch assembly for synthetic scenes: what the reference's dataset pipeline +
``Det3DDataPreprocessor_`` hand to ``UniDet3D.loss`` (formatting.py:110-142,
data_preprocessor.py:30-42,78; ``PointDetClassMappingScanNet`` transforms_3d.py:148-228 for the
GT superpoint masks).  Data plumbing only -- not on the timed path."""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .structures import Det3DDataSample, InstanceData_, PointSegData
from .synthetic import Scene


def gt_sp_masks(instance_mask: np.ndarray, superpoints: np.ndarray, n_inst: int) -> np.ndarray:
    """sp_masks[j, s] = (fraction of superpoint s's points that belong to instance j) > 0.5."""
    S = int(superpoints.max()) + 1
    cnt = np.bincount(superpoints, minlength=S).astype(np.float64)
    out = np.zeros((n_inst, S), dtype=bool)
    for j in range(n_inst):
        hit = np.bincount(superpoints[instance_mask == j], minlength=S).astype(np.float64)
        out[j] = hit / np.maximum(cnt, 1) > 0.5
    return out


def make_batch_inputs(scenes: List[Scene], device):
    """-> (batch_inputs_dict, batch_data_samples) with every tensor resident on ``device``."""
    pts, samples = [], []
    for sc in scenes:
        pts.append(torch.from_numpy(sc.points).to(device))
        n_inst = len(sc.labels)
        inst = InstanceData_(labels_3d=torch.from_numpy(sc.labels).to(device),
                             sp_masks=torch.from_numpy(gt_sp_masks(sc.instance_mask, sc.superpoints, n_inst)).to(device))
        seg = PointSegData(pts_instance_mask=torch.from_numpy(sc.instance_mask).to(device),
                           sp_pts_mask=torch.from_numpy(sc.superpoints).to(device))
        ds = Det3DDataSample(sc.lidar_path, seg, inst)
        ds.n_superpoints = int(sc.superpoints.max()) + 1
        samples.append(ds)
    return dict(points=pts), samples

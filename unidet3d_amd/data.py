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


def scene_boxes(sc: Scene):
    """Axis-aligned GT boxes (centre, size) [n_inst, 6] of a synthetic scene in its ORIGINAL frame (what a dataset with box
    annotations stores); instances without points are dropped together with their labels (second return: the kept indices)."""
    xyz = sc.points[:, :3]
    boxes, keep = [], []
    for j in range(len(sc.labels)):
        m = sc.instance_mask == j
        if m.any():
            lo, hi = xyz[m].min(0), xyz[m].max(0)
            boxes.append(np.concatenate(((lo + hi) / 2, hi - lo)))
            keep.append(j)
    return np.stack(boxes).astype(np.float32), np.asarray(keep)


def make_joint_batch(cfg: dict, scene_specs, device, seed0: int = 200, yaw_seed: int = 4):
    """Synthetic MIXED batch for the joint six-dataset config (BASELINE.json configs[3]), one scene per ``(dataset name, n_points)``
    or ``(dataset name, n_points, area_scale)`` entry of ``scene_specs``, following each dataset's annotation style
    (configs/unidet3d_1xb8_scannet_s3dis_multiscan_3rscan_scannetpp_arkitscenes.py:36-43): ScanNet / S3DIS carry instance masks
    (bbox_by_mask), the others boxes -- ARKitScenes with a heading -- and get their masks by distance in ``loss``.
    -> (scenes, dataset names, gt_boxes per scene (None or (boxes, labels)), batch_inputs_dict, batch_data_samples)."""
    from .structures import DepthInstance3DBoxes
    from .synthetic import make_scene
    dec = cfg['decoder']
    scenes, names, gt_boxes = [], [], []
    rng = np.random.default_rng(yaw_seed)
    for i, spec in enumerate(scene_specs):
        name, n_pts = spec[0], spec[1]
        d = dec['datasets'].index(name)
        sc = make_scene(seed0 + i, n_points=n_pts, n_classes=len(dec['datasets_classes'][d]), dataset=name,
                        **(dict(area_scale=float(spec[2])) if len(spec) > 2 else {}))
        scenes.append(sc)
        names.append(name)
        if cfg['bbox_by_mask'][d]:
            gt_boxes.append(None)
        else:
            b, keep = scene_boxes(sc)
            if dec['angles'][d]:
                b = np.concatenate((b, rng.uniform(-0.6, 0.6, (len(b), 1)).astype(np.float32)), 1)
            gt_boxes.append((b, sc.labels[keep]))
    inputs, samples = make_batch_inputs(scenes, device)
    for ds, gb in zip(samples, gt_boxes):
        if gb is not None:
            b, lab = gb
            ds.gt_instances_3d.labels_3d = torch.from_numpy(lab).to(device)
            ds.gt_instances_3d.sp_masks = ds.gt_instances_3d.sp_masks[:len(lab)]        # replaced by get_targets in loss()
            ds.gt_instances_3d.bboxes_3d = DepthInstance3DBoxes(torch.from_numpy(b), with_yaw=b.shape[1] == 7, box_dim=b.shape[1],
                                                                origin=(0.5, 0.5, 0.5)).to(device)
    return scenes, names, gt_boxes, inputs, samples

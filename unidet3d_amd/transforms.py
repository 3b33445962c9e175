"""Training-input transforms (host side, numpy) -- SURVEY.md section 8f rank 3.

Same names, constructor arguments and dictionary keys as the reference's pipeline steps so that a pipeline list written
for the reference drives these unchanged:

  ``load_scene_bins``            on-disk contract: ``points/*.bin`` float32 [N, 6] (xyz, rgb 0..255), ``super_points/*.bin``
                                 int64 [N], ``instance_mask/*.bin`` / ``semantic_mask/*.bin`` int64 [N]
                                 (unidet3d/loading.py:23-52, tools/scannet_data_utils.py:184-236)
  ``NormalizePointsColor_``      unidet3d/loading.py:71-107
  ``ElasticTransfrom``           unidet3d/transforms_3d.py:12-83  (adds ``elastic_coords`` = the collate input of
                                 unidet3d/unidet3d.py:162-166; the reference spells the class this way)
  ``PointDetClassMappingScanNet``   unidet3d/transforms_3d.py:148-228  (GT labels + superpoint masks)
  ``PointDetClassMappingS3DIS``     unidet3d/transforms_3d.py:86-146
  ``PointSample_``               unidet3d/transforms_3d.py:231-295

Random draws come from numpy's global generator in the reference's order (``np.random.rand`` then three ``randn`` grids
per elastic pass; ``np.random.choice`` for sampling), so a seeded run reproduces the reference's augmentation.  The noise
blur and the trilinear lookup are written as whole-array numpy expressions (no scipy objects per call); against the
reference (scipy ``convolve`` + ``RegularGridInterpolator``) the elastic coordinates agree to ~1e-5 voxel
(tests/test_ref_golden_cpu.py).

``input_dict['points']`` may be a float array / tensor [N, >=3] or any object exposing ``.tensor``.
"""
from __future__ import annotations

import numpy as np
import torch

from .registry import TRANSFORMS


def _points_array(p) -> np.ndarray:
    if hasattr(p, 'tensor'):
        p = p.tensor
    return p.detach().cpu().numpy() if torch.is_tensor(p) else np.asarray(p)


def load_scene_bins(points_path, super_points_path=None, instance_mask_path=None, semantic_mask_path=None) -> dict:
    """Reads one pre-processed scene from the reference's on-disk layout."""
    out = dict(points=np.fromfile(points_path, dtype=np.float32).reshape(-1, 6), lidar_path=str(points_path))
    for key, path in (('sp_pts_mask', super_points_path), ('pts_instance_mask', instance_mask_path),
                      ('pts_semantic_mask', semantic_mask_path)):
        if path is not None:
            out[key] = np.fromfile(path, dtype=np.int64)
            assert out[key].shape[0] == out['points'].shape[0], f'{key}: one id per point expected'
    return out


class _Transform:
    def __call__(self, input_dict):
        return self.transform(input_dict)


@TRANSFORMS.register_module()
class NormalizePointsColor_(_Transform):
    def __init__(self, color_mean, color_std=127.5):
        self.color_mean, self.color_std = color_mean, color_std

    def transform(self, input_dict):
        pts = _points_array(input_dict['points']).astype(np.float32, copy=True)
        if self.color_mean is not None:
            pts[:, 3:6] = pts[:, 3:6] - np.asarray(self.color_mean, dtype=np.float32)
        if self.color_std is not None:
            pts[:, 3:6] = pts[:, 3:6] / np.asarray(self.color_std, dtype=np.float32)
        input_dict['points'] = pts
        return input_dict


def _box_blur3(n: np.ndarray, axis: int) -> np.ndarray:
    """Mean of the 3-neighbourhood along ``axis`` with zero padding (float32, like the reference's 1/3-weight kernels)."""
    pad = [(0, 0)] * 3
    pad[axis] = (1, 1)
    p = np.pad(n, pad)
    sl = [slice(None)] * 3

    def cut(a, b):
        s = list(sl)
        s[axis] = slice(a, b)
        return p[tuple(s)]
    third = np.float32(1.0 / 3.0)
    return (cut(0, -2) * third + cut(1, -1) * third + cut(2, None) * third).astype(np.float32)


def elastic_noise_grids(extent: np.ndarray, gran: float):
    """Three blurred Gaussian noise grids for one elastic pass; draws 3 x randn(noise_dim) from numpy's global generator."""
    noise_dim = np.abs(extent).astype(np.int32) // gran + 3
    noise = [np.random.randn(noise_dim[0], noise_dim[1], noise_dim[2]).astype('float32') for _ in range(3)]
    for axis in (0, 1, 2, 0, 1, 2):
        noise = [_box_blur3(n, axis) for n in noise]
    return noise, noise_dim


def trilinear_lookup(grids, noise_dim, gran: float, x: np.ndarray) -> np.ndarray:
    """[N, 3] values of the three grids at points x; grid node i of axis d sits at (2 i - (b_d - 1)) * gran; points outside
    the node range get 0 (``bounds_error=0, fill_value=0`` of the reference's interpolator)."""
    b = np.asarray(noise_dim, dtype=np.float64)
    t = (x.astype(np.float64) + (b - 1) * gran) / (2.0 * gran)               # fractional node index
    inside = ((t >= 0) & (t <= b - 1)).all(1)
    i0 = np.clip(np.floor(t).astype(np.int64), 0, (noise_dim - 2).clip(min=0))
    f = t - i0
    out = np.zeros((x.shape[0], 3), dtype=np.float64)
    for c, g in enumerate(grids):
        g = g.astype(np.float64)
        acc = 0.0
        for dx in (0, 1):
            wx = f[:, 0] if dx else 1 - f[:, 0]
            for dy in (0, 1):
                wy = f[:, 1] if dy else 1 - f[:, 1]
                for dz in (0, 1):
                    wz = f[:, 2] if dz else 1 - f[:, 2]
                    acc = acc + g[i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz] * (wx * wy * wz)
        out[:, c] = np.where(inside, acc, 0.0)
    return out


@TRANSFORMS.register_module()
class ElasticTransfrom(_Transform):
    def __init__(self, gran, mag, voxel_size, p=1.0):
        self.gran, self.mag, self.voxel_size, self.p = gran, mag, voxel_size, p

    def transform(self, input_dict):
        coords = _points_array(input_dict['points'])[:, :3] / self.voxel_size
        if np.random.rand() < self.p:
            coords = self.elastic(coords, self.gran[0], self.mag[0])
            coords = self.elastic(coords, self.gran[1], self.mag[1])
        input_dict['elastic_coords'] = coords
        return input_dict

    def elastic(self, x, gran, mag):
        grids, noise_dim = elastic_noise_grids(np.abs(x).max(0), gran)
        return x + trilinear_lookup(grids, noise_dim, gran, x) * mag


def _sp_masks(inst: np.ndarray, n_inst: int, sp: np.ndarray) -> np.ndarray:
    """[n_inst, S] bool: more than half of a superpoint's points belong to the instance (scatter_mean(one_hot) > 0.5)."""
    S = int(sp.max()) + 1
    sel = inst >= 0
    hits = np.bincount(inst[sel] * S + sp[sel], minlength=n_inst * S).reshape(n_inst, S)
    cnt = np.bincount(sp, minlength=S)
    return 2 * hits > cnt[None]


@TRANSFORMS.register_module()
class PointDetClassMappingScanNet(_Transform):
    def __init__(self, num_classes, stuff_classes):
        self.num_classes, self.stuff_classes = num_classes, stuff_classes

    def transform(self, input_dict):
        inst = np.asarray(input_dict['pts_instance_mask']).astype(np.int64, copy=True)
        sem = np.asarray(input_dict['pts_semantic_mask']).astype(np.int64)
        sp = np.asarray(input_dict['sp_pts_mask']).astype(np.int64)
        inst[np.isin(sem, [self.num_classes] + list(self.stuff_classes))] = -1
        idxs, first, new = np.unique(inst, return_index=True, return_inverse=True)
        assert idxs[0] == -1, 'the scene must contain stuff / unlabeled points'
        inst = new.astype(np.int64) - 1                                        # contiguous ids, -1 stays -1
        n_inst = len(idxs) - 1
        input_dict['pts_instance_mask'] = inst
        input_dict['gt_labels_3d'] = sem[first[1:]] - len(self.stuff_classes)  # label of each instance's first point
        input_dict['gt_sp_masks'] = torch.from_numpy(_sp_masks(inst, n_inst, sp)) if n_inst else \
            torch.zeros((0, int(sp.max()) + 1), dtype=torch.bool)
        return input_dict


@TRANSFORMS.register_module()
class PointDetClassMappingS3DIS(_Transform):
    def __init__(self, classes):
        self.classes = classes

    def transform(self, input_dict):
        inst = np.asarray(input_dict['pts_instance_mask']).astype(np.int64, copy=True)
        sem = np.asarray(input_dict['pts_semantic_mask']).astype(np.int64)
        sp = np.asarray(input_dict['sp_pts_mask']).astype(np.int64)
        idxs, first = np.unique(inst, return_index=True)
        if idxs[0] == 1:
            inst -= 1
            idxs = idxs - 1
        assert np.array_equal(idxs, np.arange(len(idxs))), 'S3DIS instance ids must be contiguous'
        labels = sem[first]
        keep = np.isin(labels, self.classes)
        remap = np.full(len(idxs), -1, dtype=np.int64)
        remap[keep] = np.arange(int(keep.sum()))
        mapping = np.zeros(max(self.classes) + 1, dtype=np.int64)
        mapping[np.asarray(self.classes)] = np.arange(len(self.classes))
        inst = remap[inst]
        input_dict['gt_labels_3d'] = torch.from_numpy(mapping[labels[keep]])
        input_dict['gt_sp_masks'] = torch.from_numpy(_sp_masks(inst, int(keep.sum()), sp))
        input_dict['pts_instance_mask'] = inst
        return input_dict


@TRANSFORMS.register_module()
class PointSample_(_Transform):
    def __init__(self, num_points, sample_range=None, replace=False):
        self.num_points = num_points

    def transform(self, input_dict):
        pts = _points_array(input_dict['points'])
        choices = np.random.choice(range(len(pts)), min(self.num_points, len(pts)))     # with replacement, like the reference
        input_dict['points'] = pts[choices]
        inst = input_dict.get('pts_instance_mask', None)
        if inst is not None:
            inst = np.asarray(inst)[choices]
            idxs, new = np.unique(inst, return_inverse=True)
            input_dict['pts_instance_mask'] = new - 1 if idxs[0] == -1 else new       # contiguous again after dropped instances
        if input_dict.get('pts_semantic_mask', None) is not None:
            input_dict['pts_semantic_mask'] = np.asarray(input_dict['pts_semantic_mask'])[choices]
        if input_dict.get('sp_pts_mask', None) is not None:
            input_dict['sp_pts_mask'] = np.unique(np.asarray(input_dict['sp_pts_mask'])[choices], return_inverse=True)[1]
        return input_dict


def to_batch_inputs(scene_dicts, device, dataset_dirs=None):
    """Pack transformed scene dicts into what ``UniDet3D.loss`` consumes (formatting.py:110-142 + data_preprocessor.py:30-42):
    ``(batch_inputs_dict, batch_data_samples)`` with every tensor on ``device``."""
    from .structures import Det3DDataSample, InstanceData_, PointSegData
    pts, els, samples = [], [], []
    for d in scene_dicts:
        pts.append(torch.as_tensor(_points_array(d['points']), dtype=torch.float32).to(device))
        if 'elastic_coords' in d:
            els.append(torch.as_tensor(np.asarray(d['elastic_coords']), dtype=torch.float32).to(device))
        inst = InstanceData_(labels_3d=torch.as_tensor(np.asarray(d['gt_labels_3d']), dtype=torch.long).to(device),
                             sp_masks=torch.as_tensor(d['gt_sp_masks']).to(device))
        seg = PointSegData(pts_instance_mask=torch.as_tensor(d['pts_instance_mask']).to(device),
                           sp_pts_mask=torch.as_tensor(d['sp_pts_mask']).to(device))
        ds = Det3DDataSample(d.get('lidar_path', 'data/scannet/points/scene.bin'), seg, inst)
        ds.n_superpoints = int(np.asarray(d['sp_pts_mask']).max()) + 1
        samples.append(ds)
    inputs = dict(points=pts)
    if len(els) == len(pts):
        inputs['elastic_coords'] = els
    return inputs, samples

"""Deterministic synthetic ScanNet-shape scenes (host side, numpy).

The reference ships no data (``.MISSING_LARGE_BLOBS``) and there is no
network, so the bench / tests drive the hot path with synthetic rooms whose
statistics follow SURVEY.md section 8(d): a hollow room (floor + 4 walls) with
box-shell furniture, area-uniform surface samples with 4 mm thickness noise,
colours already normalised to [-1, 1] (what ``NormalizePointsColor_`` of the
reference produces, loading.py:84-107), superpoints = 0.11 m surface cells
relabelled to contiguous int64 ids (the on-disk contract of
``super_points/*.bin``, tools/scannet_data_utils.py:184-236), and per-point
instance ids for the furniture (``pts_instance_mask``).

This is data plumbing only: nothing here is on the timed path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

SEED_BASE = 20240904


@dataclass
class Scene:
    points: np.ndarray           # [N, 6] float32: xyz (m), rgb in [-1, 1]
    superpoints: np.ndarray      # [N] int64, contiguous 0..S-1
    instance_mask: np.ndarray    # [N] int64, -1 = stuff, else 0..n_inst-1
    labels: np.ndarray           # [n_inst] int64 in [0, n_classes)
    dataset: str = 'scannet'

    @property
    def lidar_path(self) -> str:
        # UniDet3D.get_dataset() keys on a path component (unidet3d.py:366-369)
        return f'data/{self.dataset}/points/synthetic.bin'


def _rect(rng, origin, e1, e2, n, sid):
    u = rng.random((n, 1))
    v = rng.random((n, 1))
    pts = origin[None] + u * e1[None] + v * e2[None]
    normal = np.cross(e1, e2)
    normal = normal / (np.linalg.norm(normal) + 1e-12)
    pts = pts + rng.normal(0.0, 0.004, (n, 1)) * normal[None]
    return pts, np.full(n, sid, np.int64)


def make_scene(scene_idx: int, n_points: int = 100_000, area_scale: float = 1.0,
               n_furniture: int = 10, n_classes: int = 18,
               sp_cell: float = 0.11, dataset: str = 'scannet') -> Scene:
    """One room. ``area_scale`` multiplies the total surface (cfg5 uses 10)."""
    rng = np.random.default_rng(SEED_BASE + scene_idx)
    s = float(np.sqrt(area_scale))
    # room ~ 1.9 x 1.6 x 1.1 m at scale 1: with the furniture shells ~12 m^2 of surface,
    # which gives ~40k occupied 2 cm voxels per 100k pts
    lx = (1.75 + 0.35 * rng.random()) * s
    ly = (1.4 + 0.35 * rng.random()) * s
    lz = 1.0 + 0.15 * rng.random()
    ex, ey, ez = np.eye(3)
    rects = []   # (origin, e1, e2, instance_id)
    rects.append((np.zeros(3), ex * lx, ey * ly, -1))                    # floor
    rects.append((np.zeros(3), ex * lx, ez * lz, -1))                    # y = 0
    rects.append((ey * ly, ex * lx, ez * lz, -1))                        # y = ly
    rects.append((np.zeros(3), ey * ly, ez * lz, -1))                    # x = 0
    rects.append((ex * lx, ey * ly, ez * lz, -1))                        # x = lx
    for f in range(n_furniture):
        size = np.array([0.15 + 0.3 * rng.random(), 0.15 + 0.3 * rng.random(),
                         0.15 + 0.4 * rng.random()]) * np.array([s ** 0.5, s ** 0.5, 1.0])
        lo = np.array([rng.random() * (lx - size[0]), rng.random() * (ly - size[1]), 0.0])
        hi = lo + size
        # 5 faces of the shell (no bottom)
        rects.append((np.array([lo[0], lo[1], hi[2]]), ex * size[0], ey * size[1], f))
        rects.append((lo, ex * size[0], ez * size[2], f))
        rects.append((np.array([lo[0], hi[1], lo[2]]), ex * size[0], ez * size[2], f))
        rects.append((lo, ey * size[1], ez * size[2], f))
        rects.append((np.array([hi[0], lo[1], lo[2]]), ey * size[1], ez * size[2], f))
    areas = np.array([np.linalg.norm(np.cross(r[1], r[2])) for r in rects])
    counts = np.floor(areas / areas.sum() * n_points).astype(np.int64)
    counts[0] += n_points - counts.sum()
    xyz, sid, inst = [], [], []
    for i, ((o, e1, e2, f), n) in enumerate(zip(rects, counts)):
        p, s_ = _rect(rng, o, e1, e2, int(n), i)
        xyz.append(p)
        sid.append(s_)
        inst.append(np.full(int(n), f, np.int64))
    xyz = np.concatenate(xyz).astype(np.float32)
    sid = np.concatenate(sid)
    inst = np.concatenate(inst)
    perm = rng.permutation(n_points)
    xyz, sid, inst = xyz[perm], sid[perm], inst[perm]
    rgb = rng.uniform(-1.0, 1.0, (n_points, 3)).astype(np.float32)
    # scene is shifted off the origin like a real scan (collate subtracts min)
    xyz = xyz + rng.uniform(-1.0, 1.0, 3).astype(np.float32)
    cell = np.floor(xyz / sp_cell).astype(np.int64)
    cell -= cell.min(0)
    dims = cell.max(0) + 1
    code = ((cell[:, 0] * dims[1] + cell[:, 1]) * dims[2] + cell[:, 2]) * len(rects) + sid
    _, superpoints = np.unique(code, return_inverse=True)
    labels = rng.integers(0, n_classes, n_furniture).astype(np.int64)
    return Scene(points=np.concatenate([xyz, rgb], 1).astype(np.float32),
                 superpoints=superpoints.astype(np.int64),
                 instance_mask=inst, labels=labels, dataset=dataset)


def make_batch(batch_size: int, first_idx: int = 0, **kw) -> List[Scene]:
    return [make_scene(first_idx + i, **kw) for i in range(batch_size)]

"""Reads a reference-style config file (plain Python dicts, e.g.
configs/unidet3d_1xb8_scannet.py) and returns its ``model`` dict; also ships the ScanNet model
dict of that config so the bench does not need the reference tree at run time."""
from __future__ import annotations

CLASSES_SCANNET = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture', 'counter',
                   'desk', 'curtain', 'refrigerator', 'showercurtrain', 'toilet', 'sink', 'bathtub', 'otherfurniture']


def load_model_cfg(path: str) -> dict:
    ns: dict = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)           # configs are plain python (mmengine Config style)
    return ns['model']


def scannet_model_cfg(num_channels: int = 32, voxel_size: float = 0.02) -> dict:
    """The ``model=dict(...)`` of configs/unidet3d_1xb8_scannet.py:32-96 (values restated)."""
    diou = dict(type='UniDet3DAxisAlignedIoULoss', mode='diou', reduction='none')
    rot = dict(type='UniDet3DRotatedIoU3DLoss', mode='diou', reduction='none')
    return dict(
        type='UniDet3D', data_preprocessor=dict(type='Det3DDataPreprocessor_'), in_channels=6,
        num_channels=num_channels, voxel_size=voxel_size, min_spatial_shape=128, query_thr=3000,
        bbox_by_mask=[True], target_by_distance=[False], use_superpoints=[True], fast_nms=[True],
        backbone=dict(type='SpConvUNet', num_planes=[num_channels * (i + 1) for i in range(5)], return_blocks=True),
        decoder=dict(type='UniDet3DEncoder', num_layers=6, datasets_classes=[CLASSES_SCANNET], in_channels=num_channels,
                     d_model=256, num_heads=8, hidden_dim=1024, dropout=0.0, activation_fn='gelu',
                     datasets=['scannet'], angles=[False]),
        criterion=dict(type='UniDet3DCriterion', datasets=['scannet'], datasets_weights=[1],
                       bbox_loss_simple=dict(diou), bbox_loss_rotated=dict(rot),
                       matcher=dict(type='UniMatcher', costs=[
                           dict(type='QueryClassificationCost', weight=0.5),
                           dict(type='BboxCostJointTraining', weight=2.0, loss_simple=dict(diou), loss_rotated=dict(rot))]),
                       loss_weight=[0.5, 1.0], non_object_weight=0.1, topk=[6], iter_matcher=True),
        train_cfg=dict(topk=6),
        test_cfg=dict(low_sp_thr=0.18, up_sp_thr=0.81, topk_insts=1000, score_thr=0, iou_thr=[0.5]))


def joint_model_cfg() -> dict:
    """The ``model=dict(...)`` of configs/unidet3d_1xb8_scannet_s3dis_multiscan_3rscan_scannetpp_arkitscenes.py (six datasets, one
    7-dof head): its values as data (configs/joint_model_cfg.json, written from the reference config by
    tools/gen_golden_reference.py; tests/test_joint_config_cpu.py pins it against tests/golden/ref_joint_model_cfg.json)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs', 'joint_model_cfg.json')) as f:
        return json.load(f)


def build_model(cfg: dict):
    from .registry import MODELS
    return MODELS.build(dict(cfg))

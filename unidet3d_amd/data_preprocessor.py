"""``Det3DDataPreprocessor_`` for point-cloud batches (unidet3d/data_preprocessor.py:8-78, which is mmdet3d's
``Det3DDataPreprocessor.simple_process`` plus the ``elastic_coords`` pass-through): takes what a dataloader of packed samples
yields -- ``dict(inputs=dict(points=[...], elastic_coords=[...]), data_samples=[...])`` or a list of such per-sample dicts --
moves the tensors to the model's device and hands back ``dict(inputs=batch_inputs_dict, data_samples=batch_data_samples)``.
Image branches of the mmdet3d class (``img`` / ``imgs`` padding, voxelisation by the preprocessor) are not on this path."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .registry import MODELS


def _to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).to(device, non_blocking=True)
    if hasattr(x, 'to'):
        return x.to(device)
    return x


@MODELS.register_module()
class Det3DDataPreprocessor_(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.register_buffer('_anchor', torch.zeros(1), persistent=False)       # follows model.to(device)

    @property
    def device(self):
        return self._anchor.device

    def forward(self, data, training: bool = False):
        if isinstance(data, (list, tuple)):                 # a list of per-sample dicts -> one dict of lists (collate_data)
            keys = data[0]['inputs'].keys()
            data = dict(inputs={k: [d['inputs'][k] for d in data] for k in keys}, data_samples=[d['data_samples'] for d in data])
        inputs, samples = data['inputs'], data.get('data_samples')
        dev = self.device
        out = {}
        if 'points' in inputs:
            out['points'] = [_to_device(p, dev).float() for p in inputs['points']]
        if 'elastic_coords' in inputs:
            out['elastic_coords'] = [_to_device(e, dev).float() for e in inputs['elastic_coords']]
        if samples is not None:
            for ds in samples:
                for field in ('gt_pts_seg', 'gt_instances_3d'):
                    obj = getattr(ds, field, None)
                    if obj is not None:
                        for k, v in list(vars(obj).items()):
                            if torch.is_tensor(v) or isinstance(v, np.ndarray) or hasattr(v, 'tensor'):
                                setattr(obj, k, _to_device(v, dev))
        if dev.type == 'cuda':
            # the uploads above are asynchronous on the current stream: a consumer on another stream (UniDet3D.prefetch) waits
            # for this event before it reads the batch
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            out['ready_event'] = ev
        return dict(inputs=out, data_samples=samples)

"""UniDet3D detector glue over the gfx950 kernels.

Drop-in for the hot part of the reference's ``UniDet3D`` (unidet3d/unidet3d.py:20-473): same
registry name, constructor arguments (:59-76), parameter names (``input_conv.0.weight``,
``unet.*``, ``output_layer.0.*``, ``decoder.*``), ``collate`` (:136-176), ``extract_feat``
(:113-134), ``_select_queries`` (:182-218), ``loss`` (:277-364) and the feature / decoder part of
``predict`` (:411-462), and the NMS / superpoint-trimming post-processing (:475-650, SURVEY.md section 8f
rank 1) through ``ops.nms_multiclass`` / ``ops.trim_boxes_by_superpoints``.

Batches are lists of per-scene tensors exactly as the reference receives them from its data
preprocessor: ``batch_inputs_dict['points']`` = List[Tensor[N_i, 6]] on the device.
"""
from __future__ import annotations

import os

import contextlib
from typing import List, Optional

import warnings

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import dense, ops
from .registry import MODELS
from .sparse import SparseBatchNorm, SparseConvTensor, SparseSequential, SubMConv3d
from .structures import DepthInstance3DBoxes, InstanceData_


_OPAQUE_SEEN = set()


def _tensors_of(obj, cuda_only: bool = True):
    """Every CUDA tensor reachable from ``obj`` through lists / tuples / sets / dicts and object attributes (``__dict__`` and
    ``__slots__``), to any depth.  The walk exists to ``record_stream`` tensors that cross from the side stream to the main one:
    a tensor it missed could be recycled by the caching allocator while still in use.  Objects without ``__dict__`` / ``__slots__``
    that are not containers (datetime, Decimal, Enum members, Path, C-extension handles ...) cannot hold tensors the walk could
    reach: they are skipped as opaque leaves, with one warning per type."""
    seen, stack = set(), [obj]
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if isinstance(o, torch.Tensor):
            if o.is_cuda or not cuda_only:
                yield o
        elif o is None or isinstance(o, (str, bytes, int, float, complex, bool, type, range, slice, nn.Module, torch.device, torch.dtype,
                                         torch.Size, np.generic, np.ndarray, torch.cuda.Event, torch.cuda.Stream)):
            continue
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif isinstance(o, (list, tuple, set, frozenset)):
            stack.extend(o)
        else:
            found = False
            if hasattr(o, '__dict__'):
                stack.extend(vars(o).values())
                found = True
            for klass in type(o).__mro__:
                slots = klass.__dict__.get('__slots__', ())
                for name in ((slots,) if isinstance(slots, str) else slots):
                    found = True
                    if name not in ('__dict__', '__weakref__') and hasattr(o, name):
                        stack.append(getattr(o, name))
            if not found and not callable(o) and type(o) not in _OPAQUE_SEEN:
                _OPAQUE_SEEN.add(type(o))
                warnings.warn(f'prefetch: treating {type(o).__module__}.{type(o).__name__} as an opaque leaf (no __dict__ / __slots__ to '
                              'look for tensors in); hand tensors over in lists / dicts / attribute objects', stacklevel=2)


@MODELS.register_module()
class UniDet3D(nn.Module):
    def __init__(self, in_channels, num_channels, voxel_size, min_spatial_shape, query_thr, use_superpoints,
                 bbox_by_mask, target_by_distance, fast_nms, use_sync_bn=True, backbone=None, decoder=None,
                 criterion=None, train_cfg=None, test_cfg=None, data_preprocessor=None, init_cfg=None):
        super().__init__()
        # mmengine BaseModel surface (train_step / val_step / test_step below): the runner feeds raw dataloader batches
        self.data_preprocessor = MODELS.build(data_preprocessor) if isinstance(data_preprocessor, dict) else data_preprocessor
        if backbone is not None:
            self.unet = MODELS.build(backbone)
        self.decoder = MODELS.build(decoder)
        self.criterion = MODELS.build(criterion)
        self.voxel_size = voxel_size
        self.min_spatial_shape = min_spatial_shape
        self.query_thr = query_thr
        self.use_superpoints = use_superpoints
        self.bbox_by_mask = bbox_by_mask
        self.target_by_distance = target_by_distance
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.use_sync_bn = use_sync_bn
        self.fast_nms = fast_nms
        # 0: IEEE divide like torch's CPU kernel (the oracle); 1: x * (1/voxel_size) like torch's CUDA
        # div-by-scalar kernel, i.e. what the reference computes when it runs on a GPU.
        self.voxel_div_mode = 0
        self._init_layers(in_channels, num_channels)
        self._vb: Optional[ops.VoxelBatch] = None
        self._packs = None
        self._side_stream = None
        self._prefetched = None
        self._staged = None

    def _init_layers(self, in_channels, num_channels):          # unidet3d.py:95-111
        self.input_conv = SparseSequential(
            SubMConv3d(in_channels, num_channels, kernel_size=3, padding=1, bias=False, indice_key='subm1'))
        self.output_layer = SparseSequential(
            SparseBatchNorm(num_channels, eps=1e-4, momentum=0.1, sync=bool(self.use_sync_bn)), nn.ReLU(inplace=True))

    # ------------------------------------------------------------------ R1
    def collate(self, points: List[torch.Tensor], elastic_points: Optional[List[torch.Tensor]] = None):
        """-> (coordinates int32 [Nv,4], features [Nv,6], inverse_mapping int64 [Np], spatial_shape)."""
        vb = ops.voxelize(points, self.voxel_size, self.min_spatial_shape, elastic_points, self.voxel_div_mode)
        self._vb = vb
        return vb.coords, vb.feats, vb.inverse, torch.tensor(vb.spatial_shape)

    def _sparse_input(self, batch_size: int) -> SparseConvTensor:
        vb = self._vb
        return SparseConvTensor(vb.feats, vb.coords, vb.spatial_shape, batch_size, index=vb.index)

    # ------------------------------------------------------------------ R5-R7
    def extract_feat(self, x: SparseConvTensor, superpoints, inverse_mapping, batch_offsets):
        """input conv -> U-Net -> BN/ReLU -> mean-pool voxel features into superpoints -> split per scene.
        ``superpoints`` is either the int64 [Np] tensor of batch-global ids (reference signature) or a
        prebuilt ``ops.PoolPlan``."""
        if hasattr(self.unet, 'prepare_geometry'):
            self.unet.prepare_geometry(x)
        if self._packs is None:
            from .sparse import WeightPacks
            self._packs = WeightPacks(self)
        self._packs.refresh()                 # all convolution weights -> MFMA fragment order in one launch, when they changed
        x = self.input_conv(x)
        x, _ = self.unet(x)
        x = self.output_layer(x)
        plan = superpoints if isinstance(superpoints, ops.PoolPlan) else self._pool_plan(superpoints, batch_offsets[-1])
        pooled = ops.superpoint_pool(x.features, plan)
        return [pooled[batch_offsets[i]:batch_offsets[i + 1]] for i in range(len(batch_offsets) - 1)]

    def _pool_plan(self, superpoints: torch.Tensor, n_superpoints: int) -> ops.PoolPlan:
        if self._vb is None:
            raise RuntimeError('extract_feat needs the voxel batch of the preceding collate()')
        return ops.PoolPlan(self._vb, superpoints, int(n_superpoints))

    # ------------------------------------------------------------------ R9
    def _select_queries(self, x, gt_instances, perms=None):
        """``perms`` (optional list of index tensors) injects the random selection for parity runs;
        the reference draws torch.randperm on the CPU RNG (unidet3d.py:209)."""
        queries, sp_centers = [], []
        for i in range(len(x)):
            if len(x[i]) > self.query_thr:
                ids = perms[i] if perms is not None else torch.randperm(len(x[i]))[:self.query_thr]
                ids = ids.to(x[i].device)
                queries.append(x[i][ids])
                sp_centers.append(gt_instances[i].sp_centers[ids])
                gt_instances[i].query_masks = gt_instances[i].sp_masks[:, ids]
                gt_instances[i].sp_centers = gt_instances[i].sp_centers[ids]
            else:
                queries.append(x[i])
                sp_centers.append(gt_instances[i].sp_centers)
                gt_instances[i].query_masks = gt_instances[i].sp_masks
        return queries, sp_centers, gt_instances

    @staticmethod
    def get_bboxes_by_masks(instance_ids: torch.Tensor, n_inst: int, points: torch.Tensor):
        """Axis-aligned boxes around each instance's points (unidet3d.py:220-256), all instances at once:
        ``instance_ids`` int64 [N] in [-1, n_inst)."""
        if n_inst == 0:
            return DepthInstance3DBoxes(points.new_zeros(0, 6), with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5))
        # one masked min / max pass per instance set (no atomics: scatter_reduce serialises on ~10 addresses)
        m = (instance_ids[None, :] == torch.arange(n_inst, device=points.device)[:, None])[:, :, None]
        lo = torch.where(m, points[None], points.new_tensor(float('inf'))).amin(1)
        hi = torch.where(m, points[None], points.new_tensor(float('-inf'))).amax(1)
        return DepthInstance3DBoxes(torch.cat(((hi + lo) / 2, hi - lo), 1), with_yaw=False, box_dim=6,
                                    origin=(0.5, 0.5, 0.5))

    def get_dataset(self, lidar_path):                           # unidet3d.py:366-369
        for dataset in self.decoder.datasets:
            if dataset in lidar_path.split('/'):
                return dataset

    def get_targets(self, points, gt_bboxes, topk):              # unidet3d.py:371-409
        float_max = points.new_tensor(1e8)
        n_boxes = len(gt_bboxes)
        centers = gt_bboxes.gravity_center
        d = ((centers[None] - points[:, None]) ** 2).sum(-1)     # [n_points, n_boxes]
        kth = torch.topk(d, min(topk + 1, len(d)), largest=False, dim=0).values[-1]
        d = torch.where(d < kth.unsqueeze(0), d, float_max)
        min_values, min_ids = d.min(dim=1)
        min_inds = torch.where(min_values < float_max, min_ids, n_boxes)
        return torch.nn.functional.one_hot(min_inds, num_classes=n_boxes + 1)[:, :-1].bool().T

    # ------------------------------------------------------------------ shared front end
    def _front(self, batch_inputs_dict, batch_data_samples, training: bool):
        points = batch_inputs_dict['points']
        elastic = batch_inputs_dict.get('elastic_coords', None) if training else None     # predict() collates xyz only (:454-455)
        B = len(points)
        self.collate(points, elastic)
        vb = self._vb
        sp_list, batch_offsets, bias = [], [0], 0
        for ds in batch_data_samples:
            sp = ds.gt_pts_seg.sp_pts_mask.to(points[0].device)
            bias = bias + int(ds.n_superpoints) if hasattr(ds, 'n_superpoints') else bias + int(sp.max().item()) + 1
            batch_offsets.append(bias)
            sp_list.append(sp)
        plan = ops.PoolPlan(vb, ops.offset_ids(sp_list, batch_offsets[:-1]), bias)      # batch-global superpoint ids
        if elastic is not None and training:
            # unidet3d.py:295-299: the training frame is (elastic - scene min) * voxel_size (ElasticTransfrom always provides
            # elastic_coords in the reference's train pipeline); the mean commutes with the scaling up to rounding
            centers = ops.superpoint_centers(vb.coord_src, plan.sp_offsets, plan.sp_points, bias, vb.stats, vb.pt_offsets)
            centers = centers * self.voxel_size
        else:
            centers = ops.superpoint_centers(vb.points, plan.sp_offsets, plan.sp_points, bias,
                                             vb.stats if training else None, vb.pt_offsets if training else None)
        sp_centers = [centers[batch_offsets[i]:batch_offsets[i + 1]] for i in range(B)]
        names = [self.get_dataset(ds.lidar_path) for ds in batch_data_samples]
        return vb, plan, batch_offsets, sp_centers, names

    # ------------------------------------------------------------------ training step (unidet3d.py:277-364)
    def _prepare_train(self, batch_inputs_dict, batch_data_samples):
        """Everything of a training step that depends on the batch alone -- voxelisation, superpoint CSR and centres, ground-truth
        boxes / targets in the training frame (unidet3d.py:295-341) and the rulebooks of all U-Net levels: integer and
        geometry kernels with a handful of host read-backs, no parameters involved."""
        vb, plan, batch_offsets, sp_centers, names = self._front(batch_inputs_dict, batch_data_samples, True)
        B = len(batch_data_samples)
        sp_gt_instances = []
        dsets = [self.decoder.datasets.index(n) for n in names]
        boxes_all, box_off = None, [0]
        by_mask = [bool(self.bbox_by_mask[d]) for d in dsets]
        if any(by_mask):
            # GT boxes of every instance of the batch's `bbox_by_mask` scenes in one pass over the points (u3d_segment_minmax_xyz); the
            # points of the other scenes of a mixed batch carry id -1 ("no instance": their boxes come with the sample).  Per scene
            # this was two masked [instances x points x 3] min / max reductions (85 us each at 100 k points).
            ids = []
            for ds, m in zip(batch_data_samples, by_mask):
                pm = ds.gt_pts_seg.pts_instance_mask.to(vb.points.device)
                ids.append(pm if m else torch.full_like(pm, -1))
                box_off.append(box_off[-1] + (len(ds.gt_instances_3d.labels_3d) if m else 0))
            if box_off[-1] > 0:
                boxes_all = ops.instance_boxes(vb, ops.offset_ids(ids, box_off[:-1], keep_negative=True), box_off[-1], self.voxel_size)
        all_boxes = None
        if boxes_all is not None:
            # ONE box object for the batch (the bottom-centre round trip of DepthInstance3DBoxes is element-wise: the same bits as per
            # scene) plus the (gravity centre, size) rows the criterion reads, computed once; the per-scene objects are row views
            all_boxes = DepthInstance3DBoxes(boxes_all, with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5))
            all_boxes.cache_gt_rows()
        for i, ds in enumerate(batch_data_samples):
            inst = ds.gt_instances_3d
            dataset = dsets[i]
            if all_boxes is not None and by_mask[i]:
                inst.bboxes_3d = all_boxes[box_off[i]:box_off[i + 1]]
            elif self.bbox_by_mask[dataset]:
                if vb.coord_src is not None:
                    pts = (batch_inputs_dict['elastic_coords'][i] - vb.stats[i, :3]) * self.voxel_size
                else:
                    pts = batch_inputs_dict['points'][i][:, :3] - vb.stats[i, :3]
                ids = ds.gt_pts_seg.pts_instance_mask.to(pts.device)
                inst.bboxes_3d = self.get_bboxes_by_masks(ids, len(inst.labels_3d), pts)
            else:
                # (the reference overwrites the sample's boxes with the shifted ones, unidet3d.py:318-330 -- harmless there, every batch
                # is loaded afresh.  A caller that feeds the SAME sample again -- bench.py, a test, an overfit run -- must not see the
                # boxes walk by -scene_min per step: the shifted object remembers the box it was made from)
                b = getattr(inst.bboxes_3d, '_u3d_unshifted', inst.bboxes_3d)
                center = b.gravity_center - (vb.stats[i, :3] * self.voxel_size if vb.coord_src is not None else vb.stats[i, :3])
                inst.bboxes_3d = DepthInstance3DBoxes(torch.cat((center, b.tensor[:, 3:]), dim=1), with_yaw=b.with_yaw,
                                                      box_dim=b.tensor.shape[1], origin=(0.5, 0.5, 0.5))
                inst.bboxes_3d._u3d_unshifted = b
                inst.bboxes_3d.cache_gt_rows()             # (gravity centre, size[, heading]): read by get_targets below and by the criterion
            inst.sp_centers = sp_centers[i]
            if self.target_by_distance[dataset]:
                inst.sp_masks = self.get_targets(inst.sp_centers, inst.bboxes_3d, self.train_cfg['topk'])
            sp_gt_instances.append(inst)
        x = self._sparse_input(B)
        if hasattr(self.unet, 'prepare_geometry'):
            self.unet.prepare_geometry(x)
        return dict(vb=vb, plan=plan, batch_offsets=batch_offsets, names=names, sp_gt_instances=sp_gt_instances, x=x)

    def prefetch(self, batch_inputs_dict, batch_data_samples, ready_event=None):
        """Run ``_prepare_train`` for the NEXT batch on a side stream so that its small kernels and host read-backs overlap the
        backward pass still executing on the main stream (a training loop calls this right after ``optimizer.step()`` has
        been queued; ``loss`` picks the result up when it is handed the same two objects).  Without it every step starts
        with the GPU idle: the first read-back of the voxeliser drains the queue and ~3 ms of launch-latency-bound
        integer work follow.  The batch tensors are only read, but they must be complete on the device before the side
        stream touches them: pass ``ready_event`` (an event recorded after their upload) or put it into the batch as
        ``batch_inputs_dict['ready_event']`` (``Det3DDataPreprocessor_`` does); without either the tensors must have been
        produced before the call by work the host has already waited for, or on the side stream itself (``prefetch_step``)."""
        pts = batch_inputs_dict['points']
        if not len(pts) or pts[0].device.type != 'cuda':
            return
        dev = pts[0].device
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(dev)
        main = torch.cuda.current_stream(dev)
        if ready_event is None:
            ready_event = batch_inputs_dict.get('ready_event')
        with torch.cuda.stream(self._side_stream):
            if ready_event is not None:
                self._side_stream.wait_event(ready_event)
            prep = self._prepare_train(batch_inputs_dict, batch_data_samples)
            done = torch.cuda.Event()
            done.record(self._side_stream)
        # the tensors were allocated from the side stream's pool but will be read by main-stream kernels: tell the caching
        # allocator, or a block freed after step i could be handed to the prefetch of step i+2 while step i+1 still reads it
        for t in _tensors_of(prep):
            t.record_stream(main)
        self._prefetched = (batch_inputs_dict, batch_data_samples, prep, done)

    def loss(self, batch_inputs_dict, batch_data_samples, query_perms=None, **kwargs):
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is batch_inputs_dict and pre[1] is batch_data_samples:
            torch.cuda.current_stream().wait_event(pre[3])
            prep = pre[2]
        else:
            prep = self._prepare_train(batch_inputs_dict, batch_data_samples)
        vb, plan, batch_offsets, names = prep['vb'], prep['plan'], prep['batch_offsets'], prep['names']
        sp_gt_instances, x = prep['sp_gt_instances'], prep['x']
        self._vb = vb
        with dense.transposed_weights(self):        # every [K, N] weight copy the backward pass needs, in one launch
            feats = self.extract_feat(x, plan, vb.inverse, batch_offsets)
            queries, sp_centers_q, sp_gt_instances = self._select_queries(feats, sp_gt_instances, query_perms)
            out = self.decoder(queries, sp_centers_q, names)
        return self.criterion(out, sp_gt_instances, names)

    # ------------------------------------------------------------------ inference up to the decoder (unidet3d.py:411-462)
    def predict_raw(self, batch_inputs_dict, batch_data_samples):
        vb, plan, batch_offsets, sp_centers, names = self._front(batch_inputs_dict, batch_data_samples, False)
        x = self._sparse_input(len(batch_data_samples))
        feats = self.extract_feat(x, plan, vb.inverse, batch_offsets)
        return self.decoder(feats, sp_centers, names)

    # ------------------------------------------------------------------ post-processing (unidet3d.py:475-538)
    def predict_by_feat(self, out, plan, vb, n_sp0, dataset_name):
        """Scene 0 of the batch, as in the reference (:498-502): softmax -> top-k over (query, class) -> class-wise NMS ->
        superpoint trimming.  Returns [(DepthInstance3DBoxes, labels, scores)]."""
        cls_preds, pred_bboxes = out['cls_preds'][0], out['bboxes'][0]
        idx = self.decoder.datasets.index(dataset_name)
        scores = F.softmax(cls_preds, dim=-1)[:, :-1]
        num_classes = scores.shape[1]
        scores, topk_idx = scores.flatten(0, 1).topk(min(self.test_cfg['topk_insts'], scores.numel()), sorted=True)
        labels = topk_idx % num_classes
        pred_bboxes = pred_bboxes[torch.div(topk_idx, num_classes, rounding_mode='floor')]
        nms_bboxes, nms_scores, nms_labels = ops.nms_multiclass(pred_bboxes, scores, labels, self.test_cfg['iou_thr'][idx],
                                                                self.test_cfg['score_thr'], bool(self.fast_nms[idx]))
        if self.use_superpoints[idx]:       # trimmed boxes are axis-aligned whatever went in (:585-592)
            nms_bboxes = ops.trim_boxes_by_superpoints(vb.points, plan.sp_offsets, plan.sp_points, n_sp0, nms_bboxes,
                                                       self.test_cfg['low_sp_thr'], self.test_cfg['up_sp_thr'])
        with_yaw = nms_bboxes.shape[1] == 7   # without trimming: 7 columns (zero heading after the fast-NMS branch, :629-636)
        boxes = DepthInstance3DBoxes(nms_bboxes, with_yaw=with_yaw, box_dim=nms_bboxes.shape[1], origin=(0.5, 0.5, 0.5))
        return [(boxes, nms_labels, nms_scores)]

    def predict(self, batch_inputs_dict, batch_data_samples, **kwargs):
        """unidet3d.py:411-473 (the reference post-processes scene 0 only -- test batches hold one scene)."""
        vb, plan, batch_offsets, sp_centers, names = self._front(batch_inputs_dict, batch_data_samples, False)
        x = self._sparse_input(len(batch_data_samples))
        feats = self.extract_feat(x, plan, vb.inverse, batch_offsets)
        out = self.decoder(feats, sp_centers, names)
        results = self.predict_by_feat(out, plan, vb, batch_offsets[1] - batch_offsets[0], names[0])
        for ds, (bboxes, labels, scores) in zip(batch_data_samples, results):
            ds.pred_instances_3d = InstanceData_(bboxes_3d=bboxes, scores_3d=scores, labels_3d=labels,
                                                 points=batch_inputs_dict['points'][0])
        return batch_data_samples

    # ------------------------------------------------------------------ what an mmengine-style runner calls (BaseModel)
    @staticmethod
    def parse_losses(losses):
        """mmengine ``BaseModel.parse_losses``: (total loss = sum of the entries whose key contains 'loss', log dict)."""
        log = {k: (v.mean() if torch.is_tensor(v) else sum(t.mean() for t in v)) for k, v in losses.items()}
        total = sum(v for k, v in log.items() if 'loss' in k)
        log = dict(loss=total, **log)
        return total, log

    def _prep(self, data, training):
        return self.data_preprocessor(data, training) if self.data_preprocessor is not None else data

    def train_step(self, data, optim_wrapper):
        """One optimisation step as ``BaseModel.train_step`` performs it: preprocess, forward(mode='loss'), parse the loss
        dict, ``optim_wrapper.update_params(loss)`` (mmengine ``OptimWrapper`` / ``AmpOptimWrapper``, tools/train.py:86-99)."""
        staged, self._staged = self._staged, None
        # mmengine enters optim_wrapper.optim_context(self) around preprocessing + forward + parse_losses: AmpOptimWrapper turns
        # autocast on there, and with accumulative_counts > 1 it is where DDP's no_sync / the accumulation bookkeeping live
        ctx = optim_wrapper.optim_context(self) if hasattr(optim_wrapper, 'optim_context') else contextlib.nullcontext()
        with ctx:
            data = staged[1] if staged is not None and staged[0] is data else self._prep(data, True)
            losses = self(data['inputs'], data['data_samples'], mode='loss')
            total, log = self.parse_losses(losses)
        optim_wrapper.update_params(total)
        return log

    def invalidate_weight_packs(self):
        """For code that writes convolution weights through ``.data`` (which does not bump ``Tensor._version``; e.g. an EMA
        parameter swap) while the model is in eval mode: the MFMA-order weight copies are rebuilt at the next forward."""
        if self._packs is not None:
            self._packs.invalidate()

    def prefetch_step(self, data):
        """``prefetch`` for loops that drive the model through ``train_step``: preprocess the NEXT raw batch (host -> device copies
        included) and queue its batch-only kernels, all on the side stream; the next ``train_step`` on the same ``data`` object
        picks both up."""
        if not torch.cuda.is_available() or next(self.parameters()).device.type != 'cuda':
            return
        dev = next(self.parameters()).device
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(dev)
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(self._side_stream):
            prepped = self._prep(data, True)
        self.prefetch(prepped['inputs'], prepped['data_samples'])          # same side stream: ordered after the uploads
        for t in _tensors_of(prepped):
            t.record_stream(main)
        self._staged = (data, prepped)

    def val_step(self, data):
        data = self._prep(data, False)
        return self(data['inputs'], data['data_samples'], mode='predict')

    test_step = val_step

    def forward(self, inputs, data_samples=None, mode='loss', **kwargs):
        if mode == 'loss':
            return self.loss(inputs, data_samples, **kwargs)
        if mode == 'predict':
            return self.predict(inputs, data_samples, **kwargs)
        if mode == 'tensor':
            return self.predict_raw(inputs, data_samples)
        raise RuntimeError(f'Invalid mode "{mode}"')

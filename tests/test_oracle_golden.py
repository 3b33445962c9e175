"""Pin the oracle's decoder restatement to golden vectors produced by the REAL
reference ``unidet3d/encoder.py`` (tools/gen_golden_encoder.py)."""
import os

import numpy as np
import torch

from _detw import fill_state_dict
from oracle import model as om
from oracle import criterion as oc

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'encoder_golden.npz'))
CLASSES = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture',
           'counter', 'desk', 'curtain', 'refrigerator', 'showercurtrain', 'toilet', 'sink',
           'bathtub', 'otherfurniture']
CLASSES_B = ['table', 'chair', 'sofa', 'bookcase', 'board']
CFG_A = dict(num_layers=6, datasets_classes=[CLASSES], in_channels=32, d_model=256, num_heads=8,
             hidden_dim=1024, dropout=0.0, activation_fn='gelu', datasets=['scannet'], angles=[False])
CFG_B = dict(num_layers=2, datasets_classes=[CLASSES, CLASSES_B], in_channels=32, d_model=256,
             num_heads=8, hidden_dim=1024, dropout=0.0, activation_fn='gelu',
             datasets=['scannet', 's3dis'], angles=[False, True])


def _close(a, b, tol=2e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    if a.size == 0:
        return
    assert np.max(np.abs(a - b)) <= tol * (1.0 + np.max(np.abs(b))), np.max(np.abs(a - b))


def test_state_dict_names_match_reference():
    m = om.OEncoder(**CFG_A)
    names = set(m.state_dict().keys())
    for k in ('input_proj.0.weight', 'input_proj.2.bias', 'self_attn_layers.0.attn.in_proj_weight',
              'self_attn_layers.5.attn.in_proj_bias', 'self_attn_layers.3.attn.out_proj.weight',
              'self_attn_layers.0.norm.weight', 'ffn_layers.0.net.0.weight', 'ffn_layers.5.net.3.bias',
              'ffn_layers.1.norm.bias', 'out_norm.weight', 'outs_cls.0.weight', 'outs_cls.2.bias',
              'out_bboxes.linear.weight'):
        assert k in names
    assert sum(p.numel() for p in m.parameters()) == 4886043      # SURVEY 2.3 (probed on the reference)


def test_decoder_forward_backward_matches_reference_golden():
    m = fill_state_dict(om.OEncoder(**CFG_A), tag0=100)
    x = [torch.from_numpy(G[f'A.x{i}']).requires_grad_() for i in range(2)]
    c = [torch.from_numpy(G[f'A.c{i}']) for i in range(2)]
    res = m(x, c, ['scannet', 'scannet'])
    loss = sum((t ** 2).sum() for t in res['cls_preds']) + sum(t.sum() for t in res['bboxes'])
    for a in res['aux_outputs']:
        loss = loss + sum((t * 0.5).sum() for t in a['cls_preds']) + sum((t ** 2).sum() for t in a['bboxes'])
    loss.backward()
    for i in range(2):
        _close(res['cls_preds'][i].detach(), G[f'A.cls{i}'])
        _close(res['bboxes'][i].detach(), G[f'A.box{i}'])
        _close(x[i].grad, G[f'A.gx{i}'], 1e-3)
        for l, a in enumerate(res['aux_outputs']):
            _close(a['cls_preds'][i].detach(), G[f'A.aux{l}.cls{i}'])
            _close(a['bboxes'][i].detach(), G[f'A.aux{l}.box{i}'])
    assert abs(loss.item() - float(G['A.loss'])) <= 1e-4 * abs(float(G['A.loss']))
    gp = dict(m.named_parameters())
    for k in G.files:
        if k.startswith('A.g.'):
            _close(gp[k[4:]].grad[:8], G[k], 1e-3)


def test_decoder_joint_datasets_and_rotated_head():
    m = fill_state_dict(om.OEncoder(**CFG_B), tag0=700)
    x = [torch.from_numpy(G[f'B.x{i}']) for i in range(3)]
    c = [torch.from_numpy(G[f'B.c{i}']) for i in range(3)]
    with torch.no_grad():
        r = m(x, c, ['s3dis', 'scannet', 's3dis'])
    for i in range(3):
        assert r['cls_preds'][i].shape == G[f'B.cls{i}'].shape
        _close(r['cls_preds'][i], G[f'B.cls{i}'])
        _close(r['bboxes'][i], G[f'B.box{i}'])
    assert r['bboxes'][0].shape[1] == 7 and r['bboxes'][1].shape[1] == 6


def test_pure_functions():
    pts = torch.from_numpy(G['F.pts']); bp = torch.from_numpy(G['F.bp'])
    _close(om.bbox_pred_to_bbox(pts, bp), G['F.box7'], 1e-6)
    _close(om.bbox_pred_to_bbox(pts, bp[:, :6]), G['F.box6'], 1e-6)
    _close(oc.bbox_to_loss(torch.from_numpy(G['F.b2l_in'])), G['F.b2l_out'], 1e-7)

"""Pin the oracle's backbone primitives (whose reference arithmetic lives in
un-vendored spconv / MinkowskiEngine / torch-scatter) to first-principles
identities with stock torch ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse_ops as so
from oracle import model as om


def _random_sparse(seed, B=2, shape=(12, 10, 9), n=150, c=5):
    g = torch.Generator().manual_seed(seed)
    pts = []
    for b in range(B):
        xyz = torch.stack([torch.randint(0, s, (n,), generator=g) for s in shape], 1)
        pts.append(torch.cat([torch.full((n, 1), b), xyz], 1))
    coords = torch.unique(torch.cat(pts), dim=0)          # lexicographic == canonical
    feats = torch.randn(len(coords), c, generator=g)
    return coords.int(), feats


def _dense(coords, feats, B, shape):
    d = torch.zeros(B, feats.shape[1], *shape)
    c = coords.long()
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = feats
    return d


@pytest.mark.parametrize('seed', [0, 1])
def test_subm_conv_equals_dense_conv3d(seed):
    shape = (12, 10, 9)
    coords, feats = _random_sparse(seed, shape=shape)
    w = torch.randn(7, 3, 3, 3, 5)
    pairs = so.build_subm_rulebook(coords, shape)
    out = so.sparse_conv(feats, w, pairs, len(coords))
    ref = F.conv3d(_dense(coords, feats, 2, shape), w.permute(0, 4, 1, 2, 3), padding=1)
    c = coords.long()
    assert torch.allclose(out, ref[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]], atol=1e-4)
    for ir, orow in pairs:                                 # canonical ordering
        assert np.all(np.diff(orow) > 0) and np.all(np.diff(ir) > 0)
    # symmetry: (i,o) in L_k  <=>  (o,i) in L_{26-k}
    for k in range(27):
        assert np.array_equal(pairs[k][0], pairs[26 - k][1])


@pytest.mark.parametrize('shape', [(12, 10, 8), (13, 11, 9)])
def test_strided_and_inverse_conv_equal_dense(shape):
    coords, feats = _random_sparse(3, shape=shape)
    w = torch.randn(6, 2, 2, 2, 5)
    oc, oshape, pairs = so.build_down_rulebook(coords, shape)
    out = so.sparse_conv(feats, w, pairs, len(oc))
    ref = F.conv3d(_dense(coords, feats, 2, shape), w.permute(0, 4, 1, 2, 3), stride=2)
    assert list(ref.shape[2:]) == [int(s) for s in oshape]
    c = oc.long()
    assert torch.allclose(out, ref[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]], atol=1e-4)
    # active outputs == cells with any active input (odd-extent edge dropped)
    occ = F.max_pool3d(_dense(coords, torch.ones(len(coords), 1), 2, shape), 2)[:, 0]
    assert int(occ.sum()) == len(oc)
    # inverse conv == conv_transpose3d sampled on the pre-downsample active set
    wi = torch.randn(4, 2, 2, 2, 6)
    up = so.sparse_conv(out, wi, pairs, len(coords), inverse=True)
    dref = F.conv_transpose3d(_dense(oc, out, 2, [int(s) for s in oshape]),
                              wi.permute(4, 0, 1, 2, 3), stride=2)
    ci = coords.long()
    inb = torch.tensor([all(int(ci[r, 1 + a]) < 2 * int(oshape[a]) for a in range(3)) for r in range(len(ci))])
    got = dref[ci[inb, 0], :, ci[inb, 1], ci[inb, 2], ci[inb, 3]]
    assert torch.allclose(up[inb], got, atol=1e-4)
    assert torch.all(up[~inb] == 0)
    for ir, orow in pairs:
        assert np.all(np.diff(ir) > 0) and np.all(np.diff(orow) > 0)


def test_voxelize_matches_unique_and_mean():
    g = torch.Generator().manual_seed(5)
    pts = [torch.rand(500, 6, generator=g) * torch.tensor([1.0, 0.8, 0.5, 1, 1, 1]) + 3.0 for _ in range(3)]
    coords, feats, inv, shape = so.voxelize(pts, 0.05, 16)
    assert list(shape) == [max(16, int(coords[:, i + 1].max()) + 1) for i in range(3)]
    # reference formulation: per-point int coords -> torch.unique(dim=0)
    allc, allf = [], []
    for b, p in enumerate(pts):
        ci = torch.floor((p[:, :3] - p[:, :3].min(0)[0]) / 0.05).long()
        allc.append(torch.cat([torch.full((len(p), 1), b), ci], 1))
        allf.append(torch.hstack((p[:, 3:], p[:, :3] - p[:, :3].mean(0))))
    allc = torch.cat(allc); allf = torch.cat(allf)
    u, uinv = torch.unique(allc, dim=0, return_inverse=True)
    assert torch.equal(u.int(), coords) and torch.equal(uinv, inv)
    s = torch.zeros(len(u), 6).index_add_(0, uinv, allf)
    cnt = torch.bincount(uinv).float()[:, None]
    assert torch.allclose(s / cnt, feats, atol=1e-6)


def test_scatter_mean_semantics():
    src = torch.randn(20, 4)
    idx = torch.tensor([0, 0, 3, 3, 3, 5] + [5] * 14)
    out = so.scatter_mean(src, idx)
    assert out.shape == (6, 4)
    assert torch.all(out[[1, 2, 4]] == 0)                  # empty segment -> 0
    assert torch.allclose(out[3], src[2:5].mean(0), atol=1e-6)


def test_unet_structure_and_names():
    m = om.OSpConvUNet([32, 64, 96, 128, 160])
    sd = m.state_dict()
    for k in ('blocks.block0.conv_branch.0.weight', 'blocks.block0.conv_branch.2.weight',
              'blocks.block1.conv_branch.5.weight', 'conv.0.running_mean', 'conv.2.weight',
              'u.blocks.block0.conv_branch.2.weight', 'deconv.0.bias', 'deconv.2.weight',
              'blocks_tail.block0.i_branch.0.weight', 'u.u.u.u.blocks.block1.conv_branch.3.weight'):
        assert k in sd, k
    assert sd['blocks_tail.block0.i_branch.0.weight'].shape == (32, 1, 1, 1, 64)
    assert sd['blocks_tail.block0.conv_branch.2.weight'].shape == (32, 3, 3, 3, 64)
    assert sd['conv.2.weight'].shape == (64, 2, 2, 2, 32)
    assert sd['deconv.2.weight'].shape == (32, 2, 2, 2, 64)
    conv_params = sum(v.numel() for k, v in sd.items() if v.dim() == 5)
    assert conv_params == 10234944 - 27 * 6 * 32 + 61440 + 655360   # SURVEY 2.3 model size facts


def test_oracle_detector_runs_cfg1_small():
    from unidet3d_amd.synthetic import make_scene
    sc = make_scene(0, n_points=3000)
    det = om.ODetector(backbone=dict(num_planes=[32, 64, 96, 128, 160]),
                       decoder=dict(num_layers=1, datasets_classes=[['a', 'b']], in_channels=32, d_model=256,
                                    num_heads=8, hidden_dim=1024, dropout=0.0, activation_fn='gelu',
                                    datasets=['scannet'], angles=[False]))
    p = [torch.from_numpy(sc.points)]
    s = [torch.from_numpy(sc.superpoints)]
    feats, x = det.extract_feat(p, s)
    assert feats[0].shape == (int(sc.superpoints.max()) + 1, 32)
    out = det.decoder(feats, det.sp_centers(p, s), ['scannet'])
    assert out['bboxes'][0].shape[1] == 6 and len(out['aux_outputs']) == 1

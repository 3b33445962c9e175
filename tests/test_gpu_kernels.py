"""GPU parity tests, kernel by kernel, through the C ABI against the CPU oracle.

Integer results (voxel coords, inverse map, rulebook pairs) must be bit-exact; floating
point within 1e-3 relative (BASELINE.json north_star), most checks far tighter.
"""
import numpy as np
import pytest
import torch

from oracle import sparse_ops as so

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _scene_points(n_scenes, n_points, seed0=0):
    from unidet3d_amd.synthetic import make_scene
    return [make_scene(seed0 + i, n_points=n_points) for i in range(n_scenes)]


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# ---------------------------------------------------------------------------- R1
@pytest.mark.parametrize('n_scenes,n_points,vs', [(1, 10_000, 0.05), (3, 30_000, 0.02)])
def test_voxelize_bit_exact(n_scenes, n_points, vs):
    from unidet3d_amd import ops
    scenes = _scene_points(n_scenes, n_points)
    pts_cpu = [torch.from_numpy(s.points) for s in scenes]
    oc, of, oinv, oshape = so.voxelize(pts_cpu, vs, 128)
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], vs, 128)
    assert vb.spatial_shape == [int(s) for s in oshape]
    assert torch.equal(vb.coords.cpu(), oc)                      # canonical order, bit exact
    assert torch.equal(vb.inverse.cpu(), oinv)
    assert _rel(vb.feats, of) < 1e-5
    # CSR of points per voxel is consistent with the inverse map
    offs = vb.vox_offsets.cpu().long(); lst = vb.vox_points.cpu().long()
    assert int(offs[-1]) == sum(len(p) for p in pts_cpu)
    rows = torch.repeat_interleave(torch.arange(len(oc)), offs[1:] - offs[:-1])
    assert torch.equal(oinv[lst], rows)


def test_voxelize_ragged_and_tiny():
    from unidet3d_amd import ops
    g = torch.Generator().manual_seed(3)
    pts_cpu = [torch.rand(1, 6, generator=g), torch.rand(777, 6, generator=g) * 3, torch.rand(5, 6, generator=g)]
    oc, of, oinv, oshape = so.voxelize(pts_cpu, 0.05, 16)
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], 0.05, 16)
    assert torch.equal(vb.coords.cpu(), oc) and torch.equal(vb.inverse.cpu(), oinv)
    assert vb.spatial_shape == [int(s) for s in oshape]
    assert _rel(vb.feats, of) < 1e-5


# ---------------------------------------------------------------------------- R2 / R3
def _check_pairs(gpu_lists, oracle_lists):
    assert len(gpu_lists) == len(oracle_lists)
    for k, ((gi, go), (oi, oo)) in enumerate(zip(gpu_lists, oracle_lists)):
        assert np.array_equal(gi, oi) and np.array_equal(go, oo), f'offset {k} differs'


@pytest.mark.parametrize('n_points,vs', [(10_000, 0.05), (60_000, 0.02)])
def test_rulebooks_bit_exact_all_levels(n_points, vs):
    from unidet3d_amd import ops, sparse
    scenes = _scene_points(2, n_points, seed0=7)
    pts_cpu = [torch.from_numpy(s.points) for s in scenes]
    oc, _, _, oshape = so.voxelize(pts_cpu, vs, 128)
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], vs, 128)
    coords, shape, index = vb.coords, vb.spatial_shape, vb.index
    for level in range(5):
        rb = sparse.build_subm_rulebook(coords, index)
        _check_pairs(rb.lists(), so.build_subm_rulebook(oc, oshape))
        if level == 4:
            break
        oc2, oshape2, opairs = so.build_down_rulebook(oc, oshape)
        c2, shape2, ix2, rb2 = sparse.build_down_rulebook(coords, 2, shape)
        assert torch.equal(c2.cpu(), oc2) and shape2 == [int(s) for s in oshape2]
        _check_pairs(rb2.lists(), opairs)
        coords, shape, index, oc, oshape = c2, shape2, ix2, oc2, oshape2


def test_index_from_raw_coords_matches_voxelizer_index():
    from unidet3d_amd import ops, sparse
    scenes = _scene_points(1, 8000, seed0=11)
    vb = ops.voxelize([torch.from_numpy(scenes[0].points).to(_dev())], 0.05, 128)
    ix = sparse.OccupancyIndex.from_coords(vb.coords, 1, vb.spatial_shape)
    assert torch.equal(ix.bitmap, vb.index.bitmap) and torch.equal(ix.rank, vb.index.rank)


# ---------------------------------------------------------------------------- K4-K8
CONV_SHAPES = [(16, 32), (32, 32), (64, 32), (64, 64), (128, 64), (96, 96), (192, 96), (128, 128), (256, 128), (160, 160)]


def _level_geometry(n_points=12_000, vs=0.05):
    from unidet3d_amd import ops, sparse
    scenes = _scene_points(2, n_points, seed0=21)
    pts_cpu = [torch.from_numpy(s.points) for s in scenes]
    oc, _, _, oshape = so.voxelize(pts_cpu, vs, 128)
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], vs, 128)
    return vb, oc, oshape


@pytest.fixture(params=['bf16x3', 'bf16x3-ts', 'bf16x3-wavetile', 'mfma'])
def math_mode(request):
    """both ways of forming fp32 products (precision.fp32_math): three exact bf16 planes per operand on the bf16 matrix pipe
    (the default: sparse convolutions through the workgroup-tile pair kernel; '-ts': SubM convolutions through the tile-stationary
    kernel of csrc/spconv_ts.hip where it is instantiated (32 / 64 channels), the rest as the default; '-wavetile': the wave-tile
    pair kernel) and the native fp32 MFMAs"""
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    with P.fp32_math(request.param.split('-')[0]), P.conv_kernel('wave' if request.param.endswith('wavetile') else 'workgroup'), \
            sparse.conv_ts(request.param.endswith('-ts')):
        yield request.param


@pytest.mark.parametrize('cin,cout', CONV_SHAPES)
def test_subm_conv_fwd_bwd(cin, cout, math_mode):
    from unidet3d_amd import sparse
    vb, oc, oshape = _level_geometry()
    n = len(oc)
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1
    add = torch.randn(n, cout, generator=g)
    go = torch.randn(n, cout, generator=g)
    xo, wo, ao = x.clone().requires_grad_(), w.clone().requires_grad_(), add.clone().requires_grad_()
    pairs = so.build_subm_rulebook(oc, oshape)
    yo = so.sparse_conv(xo, wo, pairs, n) + ao
    yo.backward(go)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    xg, wg, ag = [t.clone().to(_dev()).requires_grad_() for t in (x, w, add)]
    if cin == 16:
        xg = xg.detach()       # the (zero-padded) network input needs no gradient: dgrad 32->16 is not built
    yg = sparse.sparse_conv(xg, wg, rb, 'fwd', ag)
    yg.backward(go.to(_dev()))
    assert _rel(yg, yo) < 1e-4
    if cin != 16:
        assert _rel(xg.grad, xo.grad) < 1e-4
    assert _rel(wg.grad, wo.grad) < 1e-4
    assert _rel(ag.grad, ao.grad) < 1e-6


@pytest.mark.parametrize('T,H', [(128, 128), (128, 192), (256, 256), (64, 128)])
def test_tile_stationary_tables_bit_exact_and_conv_matches_pair_kernel(T, H):
    """csrc/spconv_ts.hip: (i) u3d_subm_halo's per-tile tables (sorted unique source rows, local positions, per-pass offset masks)
    against numpy on the rulebook's own pair lists -- integers, bit-exact; (ii) forward (+ residual addend) and input gradient of
    the tile-stationary kernel against the fp64 oracle and within fp32 rounding of the pair-list kernel, 32 / 64 channels."""
    import os
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    vb, oc, oshape = _level_geometry(n_points=20_000, vs=0.04)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    n = rb.n_out
    nbr = np.full((27, n), -1, np.int64)
    for k, (i, o) in enumerate(rb.lists()):
        nbr[k, o] = i
    nhalo, halo, loc, pmask = [t.cpu().numpy() for t in rb.halo(T, H)]
    loc, pmask = loc.view(np.uint16), pmask.view(np.uint32)
    for t in range((n + T - 1) // T):
        blk = nbr[:, t * T:(t + 1) * T]
        u = np.unique(blk[blk >= 0])
        assert nhalo[t] == len(u) and np.array_equal(halo[t, :len(u)], u), f'tile {t}: halo rows'
        ref = np.full((27, T), 0xFFFF, np.int64)
        pos = np.searchsorted(u, blk)
        ref[:, :blk.shape[1]] = np.where(blk >= 0, pos, 0xFFFF)
        assert np.array_equal(loc[t].astype(np.int64), ref), f'tile {t}: loc'
        pm = np.zeros(pmask.shape[1], np.uint32)
        kk, rr = np.nonzero(blk >= 0)
        np.bitwise_or.at(pm, pos[kk, rr] // H, np.uint32(1) << kk.astype(np.uint32))
        assert np.array_equal(pmask[t], pm), f'tile {t}: pmask'
    if T == 64:
        return                       # (tables only: the kernel is instantiated for 128- and 256-row tiles)
    pairs = so.build_subm_rulebook(oc, oshape)
    prev = os.environ.get('U3D_TS_T'), os.environ.get('U3D_TS_H')
    os.environ['U3D_TS_T'], os.environ['U3D_TS_H'] = str(T), str(H)
    try:
        for cin, cout in ((32, 32), (64, 32), (32, 64), (64, 64)):
            if sparse._ts_plan(cin, cout, n) is None or (T == 256 and (cin, cout) != (32, 32)) or (H == 256 and (cin, cout) != (32, 32)):
                continue
            g = torch.Generator().manual_seed(cin * 13 + cout)
            x, w = torch.randn(n, cin, generator=g), torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1
            add, go = torch.randn(n, cout, generator=g), torch.randn(n, cout, generator=g)
            xo = x.double().requires_grad_()
            yo = so.sparse_conv(xo, w.double(), pairs, n) + add.double()
            yo.backward(go.double())
            res = {}
            for ts in (True, False):
                with P.fp32_math('bf16x3'), sparse.conv_ts(ts):
                    xg = x.to(_dev()).requires_grad_()
                    y = sparse.sparse_conv(xg, w.to(_dev()), rb, 'fwd', add.to(_dev()))
                    y.backward(go.to(_dev()))
                    res[ts] = (y.detach(), xg.grad)
            assert _rel(res[True][0], yo) < 2e-6 and _rel(res[True][1], xo.grad) < 2e-6, (cin, cout)
            assert _rel(res[True][0], res[False][0]) < 2e-6 and _rel(res[True][1], res[False][1]) < 2e-6, (cin, cout)
    finally:
        for k, v in zip(('U3D_TS_T', 'U3D_TS_H'), prev):
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


@pytest.mark.parametrize('H', [256, 416])
def test_register_stationary_conv_matches_oracle_and_pair_kernel(H, monkeypatch):
    """csrc/spconv_ts.hip spconv_rs_k (persistent workgroups, the weights of a 32 x 32 channel block in registers, partial tiles of the
    four waves added in wave order): forward (+ residual addend) and input gradient against the fp64 oracle and within fp32 rounding of
    the pair-list kernel; 32 -> 32 and the block-tiled wider shapes; H = 256 forces tiles through more than one pass; two runs equal
    to the bit (the reduction order is fixed)."""
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    monkeypatch.setenv('U3D_RS_H', str(H))
    monkeypatch.setattr(sparse, '_RS_MIN_ROWS', 1)
    vb, oc, oshape = _level_geometry(n_points=20_000, vs=0.03)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    n = rb.n_out
    assert int(rb.halo(64, H)[0].max()) > (256 if H == 256 else 0)           # H = 256: some tile really takes a second pass
    pairs = so.build_subm_rulebook(oc, oshape)
    for cin, cout in ((32, 32), (64, 32), (32, 64), (64, 64), (96, 32)):
        g = torch.Generator().manual_seed(cin * 17 + cout)
        x, w = torch.randn(n, cin, generator=g), torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1
        add, go = torch.randn(n, cout, generator=g), torch.randn(n, cout, generator=g)
        xo = x.double().requires_grad_()
        yo = so.sparse_conv(xo, w.double(), pairs, n) + add.double()
        yo.backward(go.double())
        res = {}
        for tag, rs in (('rs', True), ('rs2', True), ('pairs', False)):
            with P.fp32_math('bf16x3'), sparse.conv_rs(rs):
                xg = x.to(_dev()).requires_grad_()
                y = sparse.sparse_conv(xg, w.to(_dev()), rb, 'fwd', add.to(_dev()))
                y.backward(go.to(_dev()))
                res[tag] = (y.detach(), xg.grad)
        assert torch.equal(res['rs'][0], res['rs2'][0]) and torch.equal(res['rs'][1], res['rs2'][1]), (cin, cout)
        assert _rel(res['rs'][0], yo) < 2e-6 and _rel(res['rs'][1], xo.grad) < 2e-6, (cin, cout)
        assert _rel(res['rs'][0], res['pairs'][0]) < 2e-6 and _rel(res['rs'][1], res['pairs'][1]) < 2e-6, (cin, cout)


@pytest.mark.parametrize('H', [256, 448])
def test_register_stationary_bf16_row_conv_matches_the_pair_kernel(H, monkeypatch):
    """spconv_rsb_k (bf16 operands gathered from bf16 rows, weights in registers, persistent workgroups): the same rounded operands as
    u3d_spconv_gmm_bf16a, fp32 accumulation in another order -- forward (+ addend) and input gradient within fp32 rounding of that
    kernel, and equal to the bit between two runs."""
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    monkeypatch.setenv('U3D_RSB_H', str(H))
    monkeypatch.setattr(sparse, '_RS_MIN_ROWS', 1)
    vb, oc, oshape = _level_geometry(n_points=20_000, vs=0.03)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    n = rb.n_out
    for cin, cout in ((32, 32), (64, 32), (32, 64), (64, 64)):
        g = torch.Generator().manual_seed(cin * 19 + cout)
        x = torch.randn(n, cin, generator=g).to(_dev())
        w = (torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1).to(_dev())
        add, go = torch.randn(n, cout, generator=g).to(_dev()), torch.randn(n, cout, generator=g).to(_dev())
        res = {}
        for tag, rs in (('rs', True), ('rs2', True), ('pairs', False)):
            with P.operands('bf16'), sparse.conv_rs_bf16(rs):
                xg = x.clone().requires_grad_()
                sparse.attach_shadow(xg, sparse.to_shadow(xg))
                gg = go.clone()
                sparse.attach_shadow(gg, sparse.to_shadow(gg))
                y = sparse.sparse_conv(xg, w, rb, 'fwd', add)
                y.backward(gg)
                res[tag] = (y.detach(), xg.grad)
        assert torch.equal(res['rs'][0], res['rs2'][0]) and torch.equal(res['rs'][1], res['rs2'][1]), (cin, cout)
        assert _rel(res['rs'][0], res['pairs'][0]) < 2e-6 and _rel(res['rs'][1], res['pairs'][1]) < 2e-6, (cin, cout)


@pytest.mark.parametrize('operands', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('tile_rows', [32, 64])
@pytest.mark.parametrize('cin,cout', [(32, 32), (64, 32), (64, 64), (96, 96), (128, 160), (256, 256)])
def test_workgroup_tile_conv_equals_wave_tile_conv_bit_for_bit(cin, cout, tile_rows, operands):
    """The workgroup-tile kernel (weights of an offset through LDS, an offset's pairs dealt evenly to four waves) performs the
    same MFMAs on the same operands in the same order per dst row as the wave-tile kernel: forward (+ residual addend) and
    input gradient must be IDENTICAL, for SubM (27 offsets) and strided / inverse (8 offsets) rulebooks, at both tile heights
    (U3D_GMM_R / U3D_GMM_G pin the plan: this small geometry would otherwise split the offsets over groups, which the
    workgroup-tile kernel is not used for)."""
    import os
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    vb, oc, oshape = _level_geometry()
    n = vb.coords.shape[0]
    g = torch.Generator().manual_seed(cin * 31 + cout)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    c2, shape2, ix2, rb2 = sparse.build_down_rulebook(vb.coords, 2, vb.spatial_shape)
    n2 = c2.shape[0]
    x = torch.randn(n, cin, generator=g).to(_dev())
    w3 = (torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1).to(_dev())
    w2 = (torch.randn(cout, 2, 2, 2, cin, generator=g) * 0.1).to(_dev())
    wi = (torch.randn(cin, 2, 2, 2, cout, generator=g) * 0.1).to(_dev())
    add = torch.randn(n, cout, generator=g).to(_dev())
    go, go2 = torch.randn(n, cout, generator=g).to(_dev()), torch.randn(n2, cout, generator=g).to(_dev())
    prev, prev_g = os.environ.get('U3D_GMM_R'), os.environ.get('U3D_GMM_G')
    os.environ['U3D_GMM_R'], os.environ['U3D_GMM_G'] = str(tile_rows), '1'
    out = {}
    try:
        for kind in ('wave', 'workgroup-all'):
            with P.conv_kernel(kind), (P.operands('bf16') if operands == 'bf16' else P.fp32_math('bf16x3')), sparse.conv_ts(False):
                xg = x.clone().requires_grad_()
                y = sparse.sparse_conv(xg, w3, rb, 'fwd', add); y.backward(go)
                xd = x.clone().requires_grad_()
                yd = sparse.sparse_conv(xd, w2, rb2, 'fwd'); yd.backward(go2)
                xu = go2.clone().requires_grad_()
                yu = sparse.sparse_conv(xu, wi, rb2, 'inv'); yu.backward(x)
                out[kind] = [y.detach(), xg.grad, yd.detach(), xd.grad, yu.detach(), xu.grad]
    finally:
        os.environ.pop('U3D_GMM_R') if prev is None else os.environ.__setitem__('U3D_GMM_R', prev)
        os.environ.pop('U3D_GMM_G') if prev_g is None else os.environ.__setitem__('U3D_GMM_G', prev_g)
    # one difference in summation ORDER: at 32-row tiles the wave-tile kernel takes 64 source channels per unit where the count
    # allows (low-order plane products of 64 channels summed before they join the running row), the workgroup-tile kernel always
    # 32 -- there the comparison is to fp32 rounding instead of bit-for-bit
    exact = operands == 'bf16' or tile_rows == 64
    for i, (name, a, b) in enumerate(zip(('subm fwd', 'subm dgrad', 'down fwd', 'down dgrad', 'inverse fwd', 'inverse dgrad'), out['wave'], out['workgroup-all'])):
        assert torch.isfinite(b).all(), name
        cs = cin if i in (0, 2, 5) else cout          # source channels of that launch
        if exact or cs % 64:
            assert torch.equal(a, b), f'{name}: max |diff| {float((a - b).abs().max())}'
        else:
            assert float((a - b).abs().max()) <= 4e-6 * float(a.abs().max()), name


def _l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize('tile_rows', [0, 64])
@pytest.mark.parametrize('cin,cout', [(32, 32), (64, 64), (128, 128)])
def test_subm_conv_bf16x3_is_as_accurate_as_the_native_fp32_mfma(cin, cout, tile_rows):
    """Forward and input gradient of a 3x3x3 submanifold convolution in both fp32 math modes against the float64 oracle, on
    inputs spanning six decades, in the maximum norm AND in the L2 norm (the latter sees an error that sits in the small
    entries): the three-plane products must not lose anything against the native fp32 MFMAs.  tile_rows = 64 forces the
    64-row wave tiles that full-size levels use (U3D_GMM_R; this geometry would plan 32)."""
    import os
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    vb, oc, oshape = _level_geometry()
    n = len(oc)
    g = torch.Generator().manual_seed(cin * 77 + cout)
    x = torch.randn(n, cin, generator=g) * torch.exp(torch.randn(n, cin, generator=g) * 2.0)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1 * torch.exp(torch.randn(cout, 3, 3, 3, cin, generator=g))
    go = torch.randn(n, cout, generator=g) * torch.exp(torch.randn(n, cout, generator=g))
    xo, wo = x.clone().double().requires_grad_(), w.clone().double().requires_grad_()
    pairs = so.build_subm_rulebook(oc, oshape)
    yo = so.sparse_conv(xo, wo, pairs, n); yo.backward(go.double())
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    err = {}
    prev = os.environ.get('U3D_GMM_R')
    if tile_rows:
        os.environ['U3D_GMM_R'] = str(tile_rows)
    try:
        for mode in ('mfma', 'bf16x3'):
            xg, wg = [t.clone().to(_dev()).requires_grad_() for t in (x, w)]
            with P.fp32_math(mode):
                yg = sparse.sparse_conv(xg, wg, rb, 'fwd'); yg.backward(go.to(_dev()))
            err[mode] = [_rel(yg, yo), _rel(xg.grad, xo.grad), _rel(wg.grad, wo.grad), _l2(yg, yo), _l2(xg.grad, xo.grad)]
    finally:
        if tile_rows:
            os.environ.pop('U3D_GMM_R') if prev is None else os.environ.__setitem__('U3D_GMM_R', prev)
    print('conv fp32 math errors (max-norm y, dx, dw; l2 y, dx) vs float64:', tile_rows, err)
    for e3, e1 in zip(err['bf16x3'], err['mfma']):
        assert e3 < max(1.5 * e1, 1e-7), err


@pytest.mark.parametrize('cin,cout', [(32, 64), (64, 96), (96, 128), (128, 160)])
def test_strided_and_inverse_conv_fwd_bwd(cin, cout, math_mode):
    from unidet3d_amd import sparse
    vb, oc, oshape = _level_geometry()
    n = len(oc)
    oc2, oshape2, pairs = so.build_down_rulebook(oc, oshape)
    c2, shape2, ix2, rb = sparse.build_down_rulebook(vb.coords, 2, vb.spatial_shape)
    n2 = len(oc2)
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, generator=g); w = torch.randn(cout, 2, 2, 2, cin, generator=g) * 0.1
    go = torch.randn(n2, cout, generator=g)
    xo, wo = x.clone().requires_grad_(), w.clone().requires_grad_()
    yo = so.sparse_conv(xo, wo, pairs, n2); yo.backward(go)
    xg, wg = x.clone().to(_dev()).requires_grad_(), w.clone().to(_dev()).requires_grad_()
    yg = sparse.sparse_conv(xg, wg, rb, 'fwd'); yg.backward(go.to(_dev()))
    assert _rel(yg, yo) < 1e-4 and _rel(xg.grad, xo.grad) < 1e-4 and _rel(wg.grad, wo.grad) < 1e-4
    # inverse conv: cout -> cin on the saved pairs
    wi = torch.randn(cin, 2, 2, 2, cout, generator=g) * 0.1
    z = torch.randn(n2, cout, generator=g); gz = torch.randn(n, cin, generator=g)
    zo, wio = z.clone().requires_grad_(), wi.clone().requires_grad_()
    uo = so.sparse_conv(zo, wio, pairs, n, inverse=True); uo.backward(gz)
    zg, wig = z.clone().to(_dev()).requires_grad_(), wi.clone().to(_dev()).requires_grad_()
    ug = sparse.sparse_conv(zg, wig, rb, 'inv'); ug.backward(gz.to(_dev()))
    assert _rel(ug, uo) < 1e-4 and _rel(zg.grad, zo.grad) < 1e-4 and _rel(wig.grad, wio.grad) < 1e-4


def test_conv_linearity_full_size():
    """Size-independent property at BASELINE cfg2 scale (one 100k-pt scene, 2 cm):
    conv(a*x + b*y) == a*conv(x) + b*conv(y), and an all-ones centre-tap kernel is the identity."""
    from unidet3d_amd import ops, sparse
    scenes = _scene_points(1, 100_000, seed0=5)
    vb = ops.voxelize([torch.from_numpy(scenes[0].points).to(_dev())], 0.02, 128)
    n = vb.coords.shape[0]
    assert 25_000 < n < 60_000
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    x = torch.randn(n, 32, device=_dev()); y = torch.randn(n, 32, device=_dev())
    w = torch.randn(32, 3, 3, 3, 32, device=_dev()) * 0.1
    lhs = sparse.sparse_conv(2.0 * x - 3.0 * y, w, rb)
    rhs = 2.0 * sparse.sparse_conv(x, w, rb) - 3.0 * sparse.sparse_conv(y, w, rb)
    assert _rel(lhs, rhs) < 1e-5
    wid = torch.zeros(32, 27, 32, device=_dev()); wid[:, 13, :] = torch.eye(32, device=_dev())
    assert torch.equal(sparse.sparse_conv(x, wid.view(32, 3, 3, 3, 32), rb), x)
    cnt = rb.counts.cpu()
    assert int(cnt[13]) == n and torch.equal(cnt, cnt.flip(0))       # centre tap hits every voxel; symmetric offsets


# ---------------------------------------------------------------------------- K9
@pytest.mark.parametrize('C', [32, 96, 160, 256])
@pytest.mark.parametrize('relu', [True, False])
def test_batchnorm_relu_fwd_bwd(C, relu):
    from unidet3d_amd import sparse
    n = 5003
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n, C, generator=g) * 2 + 0.5
    go = torch.randn(n, C, generator=g)
    bn_o = torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1)
    bn_o.weight.data = torch.rand(C, generator=g) + 0.5; bn_o.bias.data = torch.randn(C, generator=g) * 0.1
    bn_g = sparse.SparseBatchNorm(C).to(_dev())
    bn_g.load_state_dict(bn_o.state_dict())
    xo = x.clone().requires_grad_()
    yo = bn_o(xo); yo = torch.relu(yo) if relu else yo
    yo.backward(go)
    xg = x.clone().to(_dev()).requires_grad_()
    yg = bn_g(xg, relu=relu); yg.backward(go.to(_dev()))
    assert _rel(yg, yo) < 1e-5 and _rel(xg.grad, xo.grad) < 1e-4
    assert _rel(bn_g.weight.grad, bn_o.weight.grad) < 1e-4 and _rel(bn_g.bias.grad, bn_o.bias.grad) < 1e-4
    assert _rel(bn_g.running_mean, bn_o.running_mean) < 1e-5 and _rel(bn_g.running_var, bn_o.running_var) < 1e-5
    assert int(bn_g.num_batches_tracked) == int(bn_o.num_batches_tracked) == 1       # incremented inside the statistics kernel
    bn_o.eval(); bn_g.eval()
    with torch.no_grad():
        ye = bn_o(x); ye = torch.relu(ye) if relu else ye
        assert _rel(bn_g(x.to(_dev()), relu=relu), ye) < 1e-5


@pytest.mark.parametrize('cin,cout,n_points,vs', [(32, 32, 30_000, 0.02), (16, 32, 12_000, 0.05), (64, 64, 30_000, 0.02), (128, 64, 9_000, 0.05),
                                                   (128, 128, 60_000, 0.02), (96, 96, 3_000, 0.05)])
def test_batchnorm_statistics_from_the_convolution_epilogue(cin, cout, n_points, vs):
    """conv -> BN(+ReLU) with the statistics taken from the per-tile column sums the convolution kernel writes on its way out
    (no pass over the conv output) == the same pair with the norm making its own statistics pass, and == torch in float64:
    output, running statistics, every gradient.  Shapes cover 64- and 32-row tiles, several column slices, a residual addend,
    ragged last tiles and launches with offset groups (where the kernel writes no statistics and the norm falls back)."""
    from unidet3d_amd import ops, sparse
    scenes = _scene_points(2, n_points, seed0=51)
    vb = ops.voxelize([torch.from_numpy(s.points).to(_dev()) for s in scenes], vs, 128)
    n = vb.coords.shape[0]
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, generator=g); w = torch.randn(cout, 3, 3, 3, cin, generator=g) * 0.1
    add = torch.randn(n, cout, generator=g); go = torch.randn(n, cout, generator=g)
    gamma = torch.rand(cout, generator=g) + 0.5; beta = torch.randn(cout, generator=g) * 0.1
    res = {}
    for mode in (True, False):
        prev, sparse._EPILOGUE_STATS = sparse._EPILOGUE_STATS, mode
        try:
            bn = sparse.SparseBatchNorm(cout).to(_dev())
            bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
            # (the 16-channel case is the network's zero-padded 6-channel input: no input gradient exists for it)
            xd, wd, ad = x.clone().to(_dev()).requires_grad_(cin % 32 == 0), w.clone().to(_dev()).requires_grad_(), add.clone().to(_dev()).requires_grad_()
            st = {}
            f = sparse.sparse_conv(xd, wd, rb, 'fwd', ad, st)
            assert ('partial' in st) == (mode and sparse._plan(cin, cout, 27, n)[1] == 1)
            y = bn(f, relu=True, stats=st if 'partial' in st else None)
            y.backward(go.to(_dev()))
            res[mode] = dict(f=f.detach(), y=y.detach(), dx=xd.grad, dw=wd.grad, da=ad.grad, dg=bn.weight.grad, db=bn.bias.grad,
                             rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nbt=int(bn.num_batches_tracked))
        finally:
            sparse._EPILOGUE_STATS = prev
    pairs = so.build_subm_rulebook(vb.coords.cpu(), vb.spatial_shape)
    xo, wo, ao = [t.clone().double().requires_grad_() for t in (x, w, add)]
    bo = torch.nn.BatchNorm1d(cout, eps=1e-4, momentum=0.1).double()
    bo.weight.data.copy_(gamma); bo.bias.data.copy_(beta)
    fo = so.sparse_conv(xo, wo, pairs, n) + ao
    yo = torch.relu(bo(fo)); yo.backward(go.double())
    ref = dict(f=fo, y=yo, dx=xo.grad, dw=wo.grad, da=ao.grad, dg=bo.weight.grad, db=bo.bias.grad, rm=bo.running_mean, rv=bo.running_var)
    for mode in (True, False):
        for k, v in ref.items():
            if res[mode][k] is not None:
                assert _rel(res[mode][k], v) < 1e-4, (mode, k, _rel(res[mode][k], v))
        assert res[mode]['nbt'] == 1
    for k in ('y', 'dw', 'rm', 'rv'):
        assert _rel(res[True][k], res[False][k]) < 2e-6, k


def _check_levels_against_oracle(vb, pts_cpu, vs, B, n_levels=5):
    """coords / inverse / features and the SubM + strided rulebooks of ``n_levels`` levels, bit-exact against the oracle"""
    from unidet3d_amd import sparse
    oc, of, oinv, oshape = so.voxelize(pts_cpu, vs, 128)
    assert torch.equal(vb.coords.cpu(), oc) and torch.equal(vb.inverse.cpu(), oinv) and _rel(vb.feats, of) < 1e-6
    coords, shape, index = vb.coords, vb.spatial_shape, vb.index
    for level in range(n_levels):
        want = so.build_subm_rulebook(oc, oshape)
        for k, ((gi, go), (oi, oo)) in enumerate(zip(sparse.build_subm_rulebook(coords, index).lists(), want)):
            assert np.array_equal(gi, oi) and np.array_equal(go, oo), f'level {level} offset {k}'
        if level == n_levels - 1:
            break
        oc2, oshape2, opairs = so.build_down_rulebook(oc, oshape)
        c2, shape2, ix2, rb2 = sparse.build_down_rulebook(coords, B, shape)
        assert torch.equal(c2.cpu(), oc2) and shape2 == [int(x) for x in oshape2]
        for k, ((gi, go), (oi, oo)) in enumerate(zip(rb2.lists(), opairs)):
            assert np.array_equal(gi, oi) and np.array_equal(go, oo), f'level {level} down offset {k}'
        coords, shape, index, oc, oshape = c2, shape2, ix2, oc2, oshape2
    return index


def test_hashed_index_is_bit_exact(monkeypatch):
    """The hashed form of the voxel index (radix sort -> unique -> open-addressing table; csrc/hashidx.hip) forced onto an ordinary
    batch: voxel coordinates, inverse map, voxel features and the rulebooks of all five levels equal the oracle's bit for bit --
    the same kernels as with the bitmap index, fed through the table."""
    from unidet3d_amd import ops, sparse
    monkeypatch.setattr(sparse, '_INDEX_MODE', 'hash')
    scenes = _scene_points(3, 30_000, seed0=61)
    pts_cpu = [torch.from_numpy(s.points) for s in scenes]
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], 0.02, 128)
    assert vb.index.hashed
    last = _check_levels_against_oracle(vb, pts_cpu, 0.02, 3)
    assert last.hashed


def test_large_extent_scene_takes_the_hashed_index():
    """Two rooms 300 m apart at 2 cm voxels: a 15 000 x 15 000 x 128 grid, whose direct-address table would take 5.5 GB for 26 k
    voxels -- the hashed index (under 2 MB) takes over on its own; everything stays bit-exact against the oracle, and a
    convolution runs on it."""
    from unidet3d_amd import _lib as L, ops, sparse
    a, b = _scene_points(2, 25_000, seed0=71)
    pa, pb = a.points.copy(), b.points.copy()
    pb[:, 0] += 300.0; pb[:, 1] += 299.0
    pts_cpu = [torch.from_numpy(np.concatenate((pa, pb)).astype(np.float32))]
    vb = ops.voxelize([pts_cpu[0].to(_dev())], 0.02, 128)
    assert max(vb.spatial_shape) > 14_000 and L.lib().u3d_index_words(1, *vb.spatial_shape) * 12 > 4 << 30 and vb.index.hashed
    assert (vb.index.bitmap.numel() * 8 + vb.index.rank.numel() * 4) < 4 << 20
    _check_levels_against_oracle(vb, pts_cpu, 0.02, 1, n_levels=3)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    n = vb.coords.shape[0]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 32, generator=g); w = torch.randn(32, 3, 3, 3, 32, generator=g) * 0.1
    y = sparse.sparse_conv(x.to(_dev()), w.to(_dev()), rb)
    yo = so.sparse_conv(x.double(), w.double(), so.build_subm_rulebook(vb.coords.cpu(), vb.spatial_shape), n)
    assert _rel(y, yo) < 1e-5


def test_precomputed_weight_gradient_tiles_follow_the_new_row_count():
    """the tile lists a rulebook builds ahead (Rulebook.precompute_tiles, on the prefetch stream) replay what the previous
    rulebook with the same indice_key was asked for -- but a weight-gradient tile height depends on the row count, so it is recomputed
    for the new rulebook instead of replaying a height that no longer matches (a wasted launch + a second one on the main stream)."""
    from unidet3d_amd import _lib as L, ops, sparse
    sparse._TILE_HISTORY.pop(('subm', 'tkey'), None)
    rbs = []
    for seed, n_pts in ((11, 30_000), (12, 22_000)):
        vb = ops.voxelize([torch.from_numpy(s.points).to(_dev()) for s in _scene_points(2, n_pts, seed0=seed)], 0.02, 128)
        rb = sparse.build_subm_rulebook(vb.coords, vb.index)
        rb.tag = ('subm', 'tkey')
        rbs.append(rb)
    a, b = rbs
    Ta = L.lib().u3d_spconv_wgrad_tile_rows(27, a.n_out, 32, 32)
    Tb = L.lib().u3d_spconv_wgrad_tile_rows(27, b.n_out, 32, 32)
    assert Ta != Tb
    a.tile_starts('out', 64)
    ts_a = a.tile_starts('out', Ta, wgrad=(32, 32))
    b.precompute_tiles()
    assert set(b._tiles) == {('out', 64), ('out', Tb)}
    ref = sparse.Rulebook(b.pair_in, b.pair_out, b.counts, 27, b.n_in, b.n_out)._tile_starts('out', Tb)
    assert torch.equal(b._tiles[('out', Tb)], ref)
    assert ts_a.shape[0] == 27


# ---------------------------------------------------------------------------- the library's own radix sort (csrc/radix.hip)
@pytest.mark.parametrize('n,bits', [(1, 8), (63, 5), (1023, 8), (1024, 9), (1025, 17), (100_003, 17), (100_003, 40), (2_000_000, 23), (300_000, 63)])
def test_radix_sort_equals_a_stable_sort_bit_for_bit(n, bits):
    """u3d_sort_u64 (hand-written stable LSD radix sort: LDS histogram, scan, ballot-ranked scatter) against torch.sort(stable=True)
    on the CPU: keys AND permutation identical -- tile edges (1023 / 1024 / 1025 keys), heavy duplicates (5-bit keys), keys that use
    all 63 bits, 2 M keys.  The permutation being the stable one is what makes the superpoint CSR lists ascending."""
    from unidet3d_amd import _lib as L
    g = torch.Generator().manual_seed(n * 131 + bits)
    hi = (1 << bits) - 1
    keys = torch.randint(0, min(hi, (1 << 62)) + 1, (n,), generator=g, dtype=torch.int64)
    if bits == 63:
        keys = keys * 2 + torch.randint(0, 2, (n,), generator=g, dtype=torch.int64)
    if n > 10:
        keys[n // 3: n // 3 + 5] = keys[0]                    # duplicates even where keys are wide
    kd = keys.to(_dev())
    out = torch.empty_like(kd)
    perm = torch.empty(n, dtype=torch.int32, device=_dev())
    ws = L.ws(L.lib().u3d_sort_ws_bytes(n, 1), _dev())
    L.call('u3d_sort_u64', L.ptr(kd), n, bits, L.ptr(out), L.ptr(perm), L.ptr(ws), L.stream())
    ref_k, ref_p = torch.sort(keys, stable=True)
    assert torch.equal(out.cpu(), ref_k)
    assert torch.equal(perm.cpu().long(), ref_p)
    assert torch.equal(kd.cpu(), keys)                          # the input is left alone
    out2 = torch.empty_like(kd)                                 # keys only
    ws2 = L.ws(L.lib().u3d_sort_ws_bytes(n, 0), _dev())
    L.call('u3d_sort_u64', L.ptr(kd), n, bits, L.ptr(out2), None, L.ptr(ws2), L.stream())
    assert torch.equal(out2.cpu(), ref_k)


def test_csr_lists_are_ascending_inside_every_segment():
    """u3d_csr_build: offsets = counts, list = element ids stably sorted by segment -- equal to a stable argsort, segment sizes from
    0 to half of all elements (one thread sorting a segment alone would take milliseconds there), run twice: same bits."""
    from unidet3d_amd import ops
    g = torch.Generator().manual_seed(9)
    n, S = 400_000, 5000
    seg = torch.randint(0, S, (n,), generator=g)
    seg[torch.rand(n, generator=g) < 0.5] = 17                  # one huge segment (a floor superpoint)
    seg[seg == 23] = 24                                         # an empty one
    off, lst = ops.csr_build(seg.to(_dev()), S)
    ref = torch.sort(seg, stable=True)[1]
    cnt = torch.bincount(seg, minlength=S)
    assert torch.equal(off.cpu().long(), torch.cat((torch.zeros(1, dtype=torch.long), cnt.cumsum(0))))
    assert torch.equal(lst.cpu().long(), ref)
    off2, lst2 = ops.csr_build(seg.to(_dev()), S)
    assert torch.equal(lst, lst2) and torch.equal(off, off2)


# ---------------------------------------------------------------------------- K11 / K12
def test_superpoint_pool_and_centers():
    from unidet3d_amd import ops
    scenes = _scene_points(3, 20_000, seed0=31)
    pts_cpu = [torch.from_numpy(s.points) for s in scenes]
    sps, bias = [], 0
    for s in scenes:
        sp = torch.from_numpy(s.superpoints) + bias
        bias = int(sp.max()) + 1
        sps.append(sp)
    sp_all = torch.cat(sps)
    oc, of, oinv, oshape = so.voxelize(pts_cpu, 0.02, 128)
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], 0.02, 128)
    plan = ops.PoolPlan(vb, sp_all.to(_dev()), bias)
    g = torch.Generator().manual_seed(1)
    f = torch.randn(len(oc), 32, generator=g); go = torch.randn(bias, 32, generator=g)
    fo = f.clone().requires_grad_()
    po = so.scatter_mean(fo[oinv], sp_all, dim_size=bias); po.backward(go)
    fg = f.clone().to(_dev()).requires_grad_()
    pg = ops.superpoint_pool(fg, plan); pg.backward(go.to(_dev()))
    assert _rel(pg, po) < 1e-5 and _rel(fg.grad, fo.grad) < 1e-5
    cen_o = torch.cat([so.scatter_mean(p[:, :3] - p[:, :3].min(0)[0], torch.from_numpy(s.superpoints))
                       for p, s in zip(pts_cpu, scenes)])
    cen_g = ops.superpoint_centers(vb.points, plan.sp_offsets, plan.sp_points, bias, vb.stats, vb.pt_offsets)
    assert _rel(cen_g, cen_o) < 1e-5


# ---------------------------------------------------------------------------- edge cases / maximum sizes
def test_stress_1m_points_rulebook_bit_exact():
    """BASELINE cfg5 shape: one S3DIS-like room, 1M points, 2 cm voxels (~0.5M active voxels, extents > 256):
    voxel coordinates, inverse map and the level-1 / level-2 rulebooks stay bit-exact; counts fit int32."""
    from unidet3d_amd import ops, sparse
    from unidet3d_amd.synthetic import make_scene
    sc = make_scene(99, n_points=1_000_000, area_scale=10.0)
    p = [torch.from_numpy(sc.points)]
    oc, of, oinv, oshape = so.voxelize(p, 0.02, 128)
    vb = ops.voxelize([p[0].to(_dev())], 0.02, 128)
    assert len(oc) > 200_000 and max(vb.spatial_shape) > 256
    assert vb.spatial_shape == [int(s) for s in oshape]
    assert torch.equal(vb.coords.cpu(), oc) and torch.equal(vb.inverse.cpu(), oinv)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    _check_pairs(rb.lists(), so.build_subm_rulebook(oc, oshape))
    oc2, oshape2, opairs = so.build_down_rulebook(oc, oshape)
    c2, shape2, ix2, rb2 = sparse.build_down_rulebook(vb.coords, 1, vb.spatial_shape)
    assert torch.equal(c2.cpu(), oc2)
    _check_pairs(rb2.lists(), opairs)


def test_batch_with_empty_and_single_point_scenes():
    from unidet3d_amd import ops, sparse
    g = torch.Generator().manual_seed(9)
    pts_cpu = [torch.rand(300, 6, generator=g) * 2, torch.zeros(0, 6), torch.rand(1, 6, generator=g), torch.rand(50, 6, generator=g)]
    oc, of, oinv, oshape = so.voxelize([p for p in pts_cpu if len(p)], 0.05, 16)     # the oracle cannot take an empty scene
    vb = ops.voxelize([p.to(_dev()) for p in pts_cpu], 0.05, 16)
    c = vb.coords.cpu().clone()
    assert set(c[:, 0].tolist()) == {0, 2, 3}                     # scene 1 contributes no voxel
    c[:, 0] = torch.where(c[:, 0] > 1, c[:, 0] - 1, c[:, 0])      # oracle batch ids skip the empty scene
    assert torch.equal(c, oc) and torch.equal(vb.inverse.cpu(), oinv)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    x = torch.randn(len(oc), 32, device=_dev())
    w = torch.zeros(32, 27, 32, device=_dev()); w[:, 13, :] = torch.eye(32, device=_dev())
    assert torch.equal(sparse.sparse_conv(x, w.view(32, 3, 3, 3, 32), rb), x)

"""Host-side helpers added in round 5, on CPU tensors (no kernel involved): the no-copy concatenation of row views, batched id offsets,
the packed host->device copy, the per-batch ground-truth rows, the GPU test order."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cat_views_returns_the_base_only_for_a_full_consecutive_cover():
    from unidet3d_amd.ops import cat_views
    base = torch.arange(40.).reshape(10, 4).requires_grad_()
    y = base * 2                                                    # a non-leaf the views hang off (autograd must flow through it)
    parts = [y[0:3], y[3:3], y[3:7], y[7:10]]                       # incl. an empty slice
    out = cat_views(parts)
    assert out is y
    assert cat_views([y[0:3]]) is not y and torch.equal(cat_views([y[0:3]]), y[0:3])          # single element: returned as is
    for bad in ([y[0:3], y[4:10]], [y[3:7], y[0:3], y[7:10]], [y[0:5], y[5:9]], [y[:, :2][0:5], y[:, :2][5:10]], [y[0:5], (base * 3)[5:10]]):
        got = cat_views(bad)
        assert got is not y and torch.equal(got, torch.cat(bad))
    out.sum().backward()
    assert torch.equal(base.grad, torch.full_like(base, 2.0))
    assert torch.equal(cat_views([torch.ones(2, 3), torch.zeros(1, 3)]), torch.tensor([[1., 1, 1], [1, 1, 1], [0, 0, 0]]))   # plain tensors: cat


def test_cat_views_keeps_the_autograd_state_of_the_slices():
    """slices taken under no_grad of a differentiable base are detached -- torch.cat of them is too, so the (differentiable)
    base must not be returned in their place."""
    from unidet3d_amd.ops import cat_views
    y = torch.arange(24.).reshape(6, 4).requires_grad_() * 2
    with torch.no_grad():
        parts = [y[0:2], y[2:6]]
    out = cat_views(parts)
    ref = torch.cat(parts)
    assert out.requires_grad == ref.requires_grad and out.dtype == ref.dtype and torch.equal(out.detach(), ref.detach())


def test_async_weight_gradient_guard():
    """sparse.async_dw_ok: a weight gradient may stay on the side stream only for a leaf without .grad / hooks, outside
    create_graph, used for the first time in the running backward pass."""
    from unidet3d_amd import sparse
    w, b = torch.zeros(3, requires_grad=True), torch.zeros(3, requires_grad=True)
    nonleaf = w * 1.0
    sparse._ASYNC_DW_SEEN.clear()
    with torch.no_grad():                      # what a backward pass without create_graph looks like
        assert sparse.async_dw_ok(w, b)
        assert not sparse.async_dw_ok(w)       # second use in the same pass: autograd will sum the two gradients
        assert not sparse.async_dw_ok(b, None)
        sparse.join_wgrad_stream()             # a join in the middle of the pass does not forget the uses
        assert not sparse.async_dw_ok(w)
        sparse.end_of_backward()               # end of the pass: forgotten
        assert sparse.async_dw_ok(w, None)
        sparse.end_of_backward()
        w.grad = torch.zeros(3)
        assert not sparse.async_dw_ok(w)       # accumulation into an existing .grad
        w.grad = None
        h = w.register_hook(lambda g: g)
        assert not sparse.async_dw_ok(w)       # a hook reads the gradient at once
        h.remove()
        assert not sparse.async_dw_ok(nonleaf)      # not a leaf
    assert not sparse.async_dw_ok(b)           # grad mode on inside backward = create_graph
    sparse._ASYNC_DW_SEEN.clear()


def test_offset_ids_matches_the_per_scene_loop():
    from unidet3d_amd.ops import offset_ids
    g = torch.Generator().manual_seed(3)
    ids = [torch.randint(-1, 5, (n,), generator=g) for n in (7, 1, 12)]
    biases = [0, 5, 9]
    ref_keep = torch.cat([torch.where(t >= 0, t + b, t) for t, b in zip(ids, biases)])
    ref_plain = torch.cat([t + b for t, b in zip(ids, biases)])
    assert torch.equal(offset_ids(ids, biases, keep_negative=True), ref_keep)
    assert torch.equal(offset_ids(ids, biases), ref_plain)
    assert torch.equal(offset_ids([ids[0]], [4], keep_negative=True), torch.where(ids[0] >= 0, ids[0] + 4, ids[0]))
    assert all(torch.equal(a, b) for a, b in zip(ids, [t.clone() for t in ids]))              # inputs untouched


def test_h2d_pack_round_trips_mixed_dtypes_and_empty_lists():
    from unidet3d_amd._lib import h2d_pack
    specs = [([0, 3, 7], torch.int32), ([1, 2, 3, 4, 5], torch.int64), ([[1, 2, 3, 4], [5, 6, 7, 8]], torch.int32), ([0.5, 1.5], torch.float32),
             ([], torch.int32), ([9], torch.int64)]
    out = h2d_pack(specs, 'cpu')
    assert [o.dtype for o in out] == [d for _, d in specs]
    for o, (v, d) in zip(out, specs):
        assert torch.equal(o, torch.tensor(v, dtype=d).reshape(-1))
        assert o.numel() == 0 or o.data_ptr() % 16 == 0


def test_batch_box_object_hands_row_views_of_its_cached_rows():
    from unidet3d_amd.criterion import _gt_boxes
    from unidet3d_amd.structures import DepthInstance3DBoxes
    g = torch.Generator().manual_seed(5)
    raw = torch.rand(9, 6, generator=g) + 0.5
    whole = DepthInstance3DBoxes(raw, with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5))
    whole.cache_gt_rows()
    per_scene = [DepthInstance3DBoxes(raw[a:b], with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5)) for a, b in ((0, 4), (4, 4), (4, 9))]
    for (a, b), ref in zip(((0, 4), (4, 4), (4, 9)), per_scene):
        part = whole[a:b]
        assert torch.equal(part.tensor, ref.tensor)                                          # element-wise round trip: same bits batched or not
        assert part.gt_rows is not None and part.gt_rows.data_ptr() == whole.gt_rows[a:b].data_ptr()
        assert torch.equal(_gt_boxes(part), _gt_boxes(ref))
    assert whole[torch.tensor([1, 3])].gt_rows is None                                       # fancy indexing: no stale cache
    assert whole.to('cpu').gt_rows is None


def test_gpu_files_are_collected_parity_first_infrastructure_last():
    import conftest
    class It:                                                    # noqa: E306
        def __init__(self, f): self.fspath = f
    files = ['tests/test_gpu_dist.py', 'tests/test_gpu_bf16.py', 'tests/test_cabi.py', 'tests/test_gpu_eval.py', 'tests/test_gpu_kernels.py',
             'tests/test_gpu_model.py', 'tests/test_gpu_postproc.py', 'tests/test_gpu_ref_golden.py', 'tests/test_gpu_full_size.py',
             'tests/test_gpu_gradients.py']
    order = [os.path.basename(i.fspath) for i in sorted((It(f) for f in files), key=conftest._file_rank)]
    assert order == ['test_gpu_kernels.py', 'test_gpu_model.py', 'test_gpu_ref_golden.py', 'test_gpu_full_size.py', 'test_gpu_gradients.py',
                     'test_gpu_postproc.py', 'test_gpu_eval.py', 'test_cabi.py', 'test_gpu_bf16.py', 'test_gpu_dist.py']


def test_lazy_columns_compute_each_entry_once_and_slices_share_the_cache():
    """encoder._LazyColumns (the per-scene class-column selection of a mixed batch): nothing is computed until an entry is read, every
    entry is computed once, slices / negative indices / iteration / zip behave like a list's."""
    from unidet3d_amd.encoder import _LazyColumns
    calls = []
    cols = _LazyColumns([(lambda i=i: (calls.append(i), torch.tensor([i]))[1]) for i in range(6)])
    part = cols[2:4]
    assert len(cols) == 6 and len(part) == 2 and calls == []
    assert int(part[0]) == 2 and calls == [2]
    assert [int(t) for t in part] == [2, 3] and calls == [2, 3]
    assert int(cols[2]) == 2 and calls == [2, 3]                      # the slice filled the parent's cache
    assert [int(t) for t in cols] == list(range(6)) and calls == [2, 3, 0, 1, 4, 5]
    assert int(cols[-1]) == 5 and [int(t) for t in cols[4:][::-1]] == [5, 4]
    assert [(a, int(t)) for a, t in zip('ab', part)] == [('a', 2), ('b', 3)]


def test_gravity_center_forms_agree_bit_for_bit():
    """DepthInstance3DBoxes.gravity_center: the three-launch form, the cached GT rows and the reference's z + h / 2 give the same bits;
    row slices carry the matching cached rows."""
    from unidet3d_amd.structures import DepthInstance3DBoxes
    g = torch.Generator().manual_seed(0)
    t = torch.randn(13, 7, generator=g)
    t[:, 3:6] = t[:, 3:6].abs() + 0.1
    b = DepthInstance3DBoxes(t, with_yaw=True, box_dim=7, origin=(0.5, 0.5, 0.5))
    want = b.tensor[:, :3].clone()
    want[:, 2] = b.tensor[:, 2] + b.tensor[:, 5] * 0.5
    assert torch.equal(b.gravity_center, want)
    rows = b.cache_gt_rows()
    assert rows.shape == (13, 7) and torch.equal(rows[:, :3], want) and torch.equal(rows[:, 3:], b.tensor[:, 3:])
    assert torch.equal(b.gravity_center, want)                         # now served from the cache
    part = b[3:9]
    assert torch.equal(part.gravity_center, want[3:9]) and torch.equal(part.gt_rows, rows[3:9])
    assert b[torch.tensor([1, 5])].gt_rows is None and torch.equal(b[torch.tensor([1, 5])].gravity_center, want[[1, 5]])

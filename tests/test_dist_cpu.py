"""N>1 path on CPU: world_size-2 gloo processes exercise the flat-gradient all-reduce and the
SyncBatchNorm statistic exchange (the collective layer is device agnostic; the kernels feeding it
are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from unidet3d_amd.dist import FlatGradBucket, broadcast_params, init_from_env
    from unidet3d_amd.sparse import allreduce_bn_sums
    r, w, _ = init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                     # ranks start different, broadcast must align them
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    broadcast_params(net)
    bucket = FlatGradBucket(net.parameters(), attach=(rank == 0))     # rank 0: in-place views; rank 1: pack() after backward
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g); Y = torch.randn(8, 3, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    if rank == 0:
        bucket.zero()
    else:
        bucket.clear_grads()
    ((net(xs) - ys) ** 2).mean().backward()
    if rank == 1:
        bucket.pack()
    assert bucket.check_views()
    bucket.allreduce_mean()
    # the bucketed, backward-overlapped exchange must give the same averaged gradients
    net2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    net2.load_state_dict(net.state_dict())
    ov = FlatGradBucket(net2.parameters(), attach=False).enable_overlap(bucket_bytes=64)      # 3 buckets for 4 tensors
    assert len(ov.buckets) == 3
    for _ in range(2):                                # hooks re-arm for the next step
        ov.clear_grads()
        ((net2(xs) - ys) ** 2).mean().backward()
        ov.finish()
        assert ov.check_views() and torch.allclose(ov.flat, bucket.flat, atol=1e-7)
    # SyncBN statistics: per-rank partial sums of different row counts -> global mean / var
    C = 4
    xr = torch.randn(10 + 7 * rank, C, generator=torch.Generator().manual_seed(rank)).double()
    sums = torch.cat([xr.sum(0), (xr * xr).sum(0), torch.tensor([float(len(xr))], dtype=torch.float64)])
    allreduce_bn_sums(sums)
    q.put((rank, bucket.flat.clone().numpy(), [p.detach().clone().numpy() for p in net.parameters()], sums.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_and_syncbn_stats_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # identical averaged gradients on both ranks == gradient of the full-batch mean loss
    assert np.allclose(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)                   # broadcast aligned the replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    with torch.no_grad():
        for p, a in zip(net.parameters(), res[0][2]):
            p.copy_(torch.from_numpy(a))
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g); Y = torch.randn(8, 3, generator=g)
    ((net(X) - Y) ** 2).mean().backward()
    full = torch.cat([p.grad.view(-1) for p in net.parameters()]).numpy()
    assert np.allclose(res[0][1], full, atol=1e-6)
    # statistics equal those of the concatenated batch
    xs = [torch.randn(10 + 7 * r, 4, generator=torch.Generator().manual_seed(r)).double() for r in range(world)]
    allx = torch.cat(xs)
    s = res[0][3]
    n = s[8]
    assert n == len(allx)
    assert np.allclose(s[:4] / n, allx.mean(0).numpy()) and np.allclose(s[4:8] / n - (s[:4] / n) ** 2, allx.var(0, unbiased=False).numpy())


def test_overlap_buckets_with_unused_parameters_single_process():
    """A parameter that receives no gradient must not hold its bucket back: finish() closes it with zeros."""
    from unidet3d_amd.dist import FlatGradBucket
    torch.manual_seed(0)
    used, unused = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    params = list(used.parameters()) + list(unused.parameters())
    b = FlatGradBucket(params, attach=False).enable_overlap(bucket_bytes=32)
    for _ in range(2):
        b.clear_grads()
        used(torch.ones(2, 4)).sum().backward()
        b.finish()
        assert b.check_views()
        assert torch.equal(used.weight.grad, torch.full((3, 4), 2.0)) and torch.equal(used.bias.grad, torch.full((3,), 2.0))
        assert float(unused.weight.grad.abs().sum()) == 0.0 and float(unused.bias.grad.abs().sum()) == 0.0
    assert float(b.clip_grad_norm_(1.0)) > 1.0 and abs(float(b.flat.norm()) - 1.0) < 1e-5


# ---- 4 ranks, unequal scenes, one rank without ground truth, one rank that skips a parameter ---------------------------
class _Head(torch.nn.Module):
    """A stand-in decoder head: per-query class logits and (exp-sized) boxes, plus a branch only some ranks use."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Linear(8, 16)
        self.cls = torch.nn.Linear(16, 6)
        self.box = torch.nn.Linear(16, 6)
        self.extra = torch.nn.Linear(16, 16)

    def forward(self, x, use_extra):
        h = torch.relu(self.trunk(x))
        if use_extra:
            h = h + self.extra(h)
        b = self.box(h)
        return self.cls(h), torch.cat((b[:, :3], torch.exp(b[:, 3:])), 1)


def _scene_of(rank):
    """(queries [n, 8], gt boxes [g, 6], labels [g], query_masks [g, n]); rank 2 has no ground truth."""
    g = torch.Generator().manual_seed(50 + rank)
    n, gts = 20 + 7 * rank, (0 if rank == 2 else 2 + rank)
    x = torch.randn(n, 8, generator=g)
    gt = torch.cat((torch.rand(gts, 3, generator=g) * 2, torch.rand(gts, 3, generator=g) + 0.3), 1)
    return x, gt, torch.randint(0, 5, (gts,), generator=g), torch.rand(gts, n, generator=g) < 0.6


def _loss_of(net, rank):
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import scannet_model_cfg
    from unidet3d_amd.registry import MODELS
    from unidet3d_amd.structures import DepthInstance3DBoxes, InstanceData_
    crit = MODELS.build(scannet_model_cfg()['criterion'])
    x, gt, labels, qm = _scene_of(rank)
    cls, box = net(x, use_extra=rank != 1)                   # rank 1 never touches ``extra``: no gradient for it there
    inst = InstanceData_(labels_3d=labels, query_masks=qm, bboxes_3d=DepthInstance3DBoxes(gt, with_yaw=False, box_dim=6, origin=(0.5, 0.5, 0.5)))
    return crit(dict(cls_preds=[cls], bboxes=[box], aux_outputs=[]), [inst], ['scannet'])['det_loss']


def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from unidet3d_amd.dist import FlatGradBucket, broadcast_params, init_from_env
    init_from_env('gloo')
    torch.manual_seed(0)
    net = _Head()
    broadcast_params(net)
    b = FlatGradBucket(net.parameters(), attach=False).enable_overlap(bucket_bytes=256)
    assert len(b.buckets) >= 4
    order = []
    launch = b._launch
    b._launch = lambda bk, sync=False: (order.append(bk['index']), launch(bk, sync))[1]
    for _ in range(2):
        b.clear_grads()
        _loss_of(net, rank).backward()
        b.finish()
        assert b.check_views()
    q.put((rank, b.flat.clone().numpy(), order))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_order_is_rank_independent_world4_with_empty_gt_and_unused_parameter():
    """the buckets' all-reduces must start in the same order on every rank even when a rank has no ground truth
    (box loss contributes a graph-connected zero) or skips a parameter altogether; the averaged gradient equals the mean of
    the four single-process gradients."""
    world, port = 4, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_b = max(res[0][2]) + 1
    for r in res:
        assert r[2] == list(range(n_b - 1, -1, -1)) * 2            # descending bucket index, both steps, on every rank
        assert np.allclose(r[1], res[0][1])
    torch.manual_seed(0)
    net = _Head()
    want = None
    for rank in range(world):
        net.zero_grad()
        _loss_of(net, rank).backward()
        g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
        want = g if want is None else want + g
    assert np.allclose(res[0][1], (want / world).numpy(), atol=1e-6)
    assert float(np.abs(res[0][1]).max()) > 0


def _forced_worker(port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    from unidet3d_amd import dist as D
    assert D.init_from_env('gloo') == (0, 1, 0) and not dist.is_initialized()        # one rank: no group unless forced
    D.init_from_env('gloo', force=True)
    assert dist.is_initialized() and dist.get_world_size() == 1
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, *a, **kw: (calls.append((t.dtype, kw.get('group') is not None)), real(t, *a, **kw))[1]
    torch.manual_seed(3)
    base = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    out = {}
    for forced in (False, True):
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))      # (the bucket's hooks stay on the parameters)
        net.load_state_dict(base.state_dict())
        D.force_collectives(forced)
        assert D.collectives_on() == forced
        n0 = len(calls)
        b = D.FlatGradBucket(net.parameters(), attach=False).enable_overlap(bucket_bytes=64)
        assert (b.group is not None) == forced                                        # the buckets' own communicator exists only when collectives are on
        b.clear_grads()
        ((net(x) - y) ** 2).mean().backward()
        b.finish()
        out[forced] = (b.flat.clone(), len(calls) - n0)
    D.force_collectives(False)
    dist.all_reduce = real
    q.put((out[False][1], out[True][1], bool(torch.equal(out[False][0], out[True][0])), all(g for _, g in calls)))
    dist.destroy_process_group()


def test_forced_collectives_on_a_one_rank_group_are_the_identity():
    """dist.force_collectives(): the bring-up mode tests/test_gpu_dist.py uses to drive the RCCL call sequence on one GPU -- here on
    gloo: without it a one-rank group issues no collective, with it every bucket goes out (on the buckets' own communicator) and the
    reduced gradients are bit-identical."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    n_plain, n_forced, equal, own_group = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert n_plain == 0 and n_forced == 3 and equal and own_group

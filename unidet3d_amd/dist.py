"""Data-parallel harness: one process per GPU, replicated weights, one scene batch per rank.

The reference gets this from mmengine's DDP wrapper + ``nn.SyncBatchNorm``.
Here: all parameter gradients live in ONE flat fp32 buffer (``p.grad`` are views into it).  The buffer is cut
into a few contiguous buckets (default 16 MB: xGMI rings are per-link bound, a handful of large collectives beats
many small ones); a bucket is all-reduced asynchronously (RCCL over xGMI) as soon as backward has produced the last
of its gradients, so the exchange of the decoder's gradients overlaps with the backbone's backward
(``enable_overlap()`` / ``finish()``).  Batch-norm statistics are exchanged by ``sparse.allreduce_bn_sums``.
Works unchanged on CPU tensors with the gloo backend (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


_FORCE = False


def _join_side_streams():
    from .sparse import join_wgrad_stream
    join_wgrad_stream()


def force_collectives(on: bool = True) -> bool:
    """Issue every collective of the data-parallel path even when the process group has ONE rank (SyncBatchNorm sums, gradient
    buckets, the buckets' own communicator).  A single-GPU box can then drive the exact RCCL call sequence of an N-GPU run --
    communicator creation, fp64 / fp32 all-reduces, stream hand-over -- with results that must equal the non-distributed step bit
    for bit (tests/test_gpu_dist.py).  Returns the previous setting."""
    global _FORCE
    prev, _FORCE = _FORCE, bool(on)
    return prev


def collectives_on(group=None) -> bool:
    """True when the data-parallel collectives are to be issued: a process group with more than one rank, or any initialised group
    under ``force_collectives()``."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return _FORCE or dist.get_world_size(group) > 1


class FlatGradBucket:
    """One flat fp32 buffer for all parameter gradients.

    ``attach()`` makes every ``p.grad`` a view of the buffer (autograd then accumulates in place);
    ``pack()`` is the cheaper per-step form: gradients are produced by backward as fresh tensors
    (``p.grad = None`` beforehand, so no accumulate kernels), copied into the flat buffer with one
    multi-tensor copy, and ``p.grad`` is re-pointed at the views so the optimizer reads the reduced
    values."""

    def __init__(self, params: Iterable[torch.nn.Parameter], attach: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views = []
        o = 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        if attach:
            self.attach()

    def attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        self.flat.zero_()

    def clear_grads(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        _join_side_streams()
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        self.attach()

    def clip_grad_norm_(self, max_norm: float, eps: float = 1e-6) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_(norm_type=2) on the flat buffer: two kernels instead of one
        multi-tensor pass per parameter list (grads must be packed / attached)."""
        total = torch.linalg.vector_norm(self.flat, 2)
        self.flat.mul_(torch.clamp(max_norm / (total + eps), max=1.0))
        return total

    def allreduce_mean(self, group=None):
        """Sum over ranks / world size (DDP semantics).  No-op without a process group."""
        if collectives_on(group):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))

    # ---- bucketed exchange overlapped with backward ---------------------------------------
    def enable_overlap(self, bucket_bytes: int = 16 << 20, group=None):
        """Cut the flat buffer into contiguous buckets and hook every parameter: when backward has accumulated the
        last gradient of a bucket, its gradients are copied into the buffer (one multi-tensor copy) and the bucket's
        all-reduce is launched asynchronously.  Use with ``clear_grads()`` before and ``finish()`` after backward.
        Assumes every parameter receives at most one gradient per backward (true for this model); ``finish()`` raises
        if a second accumulation hit a bucket that was already launched."""
        if group is None and collectives_on():
            # a communicator of its own: the (blocking) SyncBatchNorm all-reduces of the backbone's backward would otherwise
            # queue behind a 16 MB bucket on the shared one.  Collective call: every rank reaches enable_overlap().
            group = dist.new_group()
        self.group = group
        self.buckets = []
        i0, o0, o = 0, 0, 0
        for i, p in enumerate(self.params):
            o += p.numel()
            if (o - o0) * self.flat.element_size() >= bucket_bytes or i == len(self.params) - 1:
                self.buckets.append(dict(lo=i0, hi=i + 1, flat=self.flat[o0:o], pending=i + 1 - i0, launched=False, dirty=False, work=None))
                i0, o0 = i + 1, o
        self._bucket_of = {}
        for k, b in enumerate(self.buckets):
            b['index'] = k
            for i in range(b['lo'], b['hi']):
                self._bucket_of[id(self.params[i])] = b
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)
        # Collectives on one communicator must be issued in the SAME order on every rank.  Backward produces gradients
        # from the last parameter to the first, so buckets are launched strictly in descending index order: bucket k goes
        # out only when it is complete AND every bucket above it has been launched (torch DDP's rule).  A parameter that
        # receives no gradient on one rank only therefore cannot reorder that rank's collectives; finish() flushes the
        # rest in the same order.
        self._next = len(self.buckets) - 1
        self._overlap = True
        return self

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _on_grad(self, p):
        b = self._bucket_of[id(p)]
        if b['launched']:
            b['dirty'] = True
            return
        b['pending'] -= 1
        while self._next >= 0 and self.buckets[self._next]['pending'] == 0:
            self._launch(self.buckets[self._next])
            self._next -= 1

    def _launch(self, b, sync: bool = False):
        _join_side_streams()                   # weight gradients still running on the side stream (sparse.set_wgrad_overlap(2))
        src, dst = [], []
        for i in range(b['lo'], b['hi']):
            p, v = self.params[i], self.views[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        b['launched'] = True
        if collectives_on(self.group):
            b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=not sync)

    def finish(self):
        """Call after backward: launches the buckets that never completed (unused parameters), waits for the
        collectives, averages, and re-arms the hooks' counters for the next step."""
        while self._next >= 0:                  # descending index order, like the hooks
            self._launch(self.buckets[self._next])
            self._next -= 1
        for b in self.buckets:
            if b['work'] is not None and b['work'] is not True:
                b['work'].wait()
            if b['dirty']:                       # a gradient was accumulated into a bucket that was already being reduced
                raise RuntimeError('FlatGradBucket: a parameter received a second gradient after its bucket was launched; '
                                   'use pack() + allreduce_mean() for graphs that reuse parameters')
        w = self._world()
        if w > 1 or collectives_on(self.group):
            self.flat.div_(w)
        for b in self.buckets:
            b.update(pending=b['hi'] - b['lo'], launched=False, dirty=False, work=None)
        self._next = len(self.buckets) - 1

    def sync(self):
        """After backward: the overlapped form when ``enable_overlap()`` was called, else one copy + one all-reduce."""
        if getattr(self, '_overlap', False):
            self.finish()
        else:
            self.pack()
            self.allreduce_mean()

    def check_views(self) -> bool:
        """True while every p.grad still aliases the flat buffer (autograd accumulates in place)."""
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * self.flat.element_size()
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params)


def init_from_env(backend: str = 'nccl', force: bool = False):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank).  ``force``: create the
    process group even for WORLD_SIZE = 1 (see ``force_collectives``)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def broadcast_params(module: torch.nn.Module, src: int = 0):
    if collectives_on():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)

"""Data-parallel harness: one process per GPU, replicated weights, one scene batch per rank.

The reference gets this from mmengine's DDP wrapper + ``nn.SyncBatchNorm``.
Here: all parameter gradients live in ONE flat fp32 buffer (``p.grad`` are views into it), so the
gradient exchange is a single RCCL all-reduce of 63.5 MB over xGMI with no pack/unpack copies, and
batch-norm statistics are exchanged by ``sparse.allreduce_bn_sums``.  Works unchanged on CPU tensors
with the gloo backend (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()

    def allreduce_mean(self, group=None):
        """Sum over ranks / world size (DDP semantics).  No-op without a process group."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))

    def check_views(self) -> bool:
        """True while every p.grad still aliases the flat buffer (autograd accumulates in place)."""
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * self.flat.element_size()
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params)


def init_from_env(backend: str = 'nccl'):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def broadcast_params(module: torch.nn.Module, src: int = 0):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)

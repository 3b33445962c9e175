"""Signed relative error of the NT GEMM in both fp32 math modes on all-positive operands (every product and every partial sum
positive: a truncating accumulator shows as a negative mean error, round-to-nearest as ~0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd import precision as P
from unidet3d_amd.dense import _gemm_nt
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for K in (32, 256, 2048):
    a = (torch.rand(4096, K, generator=g) + 0.5); w = (torch.rand(256, K, generator=g) + 0.5); b = torch.zeros(256)
    ref = a.double() @ w.double().t()
    line = f'K={K:5d}:'
    for mode in ('mfma', 'bf16x3'):
        with P.fp32_math(mode):
            y = _gemm_nt(a.to(dev), w.to(dev), b.to(dev)).double().cpu()
        e = (y - ref) / ref
        line += f'  {mode}: mean {e.mean():+.3e} rms {e.pow(2).mean().sqrt():.3e} max {e.abs().max():.3e}'
    yt = (a.to(dev) @ w.to(dev).t()).double().cpu(); e = (yt - ref) / ref
    print(line + f'  torch: mean {e.mean():+.3e} rms {e.pow(2).mean().sqrt():.3e}', flush=True)

# sparse convolution, all-positive features and weights
from unidet3d_amd import ops, sparse
from unidet3d_amd.synthetic import make_scene
vb = ops.voxelize([torch.from_numpy(make_scene(i, n_points=30000).points).to(dev) for i in range(2)], 0.04, 128)
rb = sparse.build_subm_rulebook(vb.coords, vb.index)
n = vb.coords.shape[0]
for C in (32, 64, 128):
    x = torch.rand(n, C, generator=g) + 0.5; w = torch.rand(C, 3, 3, 3, C, generator=g) + 0.5
    lists = rb.lists()
    ref = torch.zeros(n, C, dtype=torch.float64)
    wd = w.double().reshape(C, 27, C)
    for k, (gi, go) in enumerate(lists):
        if len(gi):
            ref.index_add_(0, torch.as_tensor(go, dtype=torch.int64), x.double()[torch.as_tensor(gi, dtype=torch.int64)] @ wd[:, k, :].t())
    line = f'conv C={C:4d} n={n}:'
    for mode in ('mfma', 'bf16x3'):
        with P.fp32_math(mode):
            y = sparse.sparse_conv(x.to(dev), w.to(dev), rb, 'fwd').double().cpu()
        e = (y - ref) / ref
        line += f'  {mode}: mean {e.mean():+.3e} rms {e.pow(2).mean().sqrt():.3e}'
    print(line, flush=True)

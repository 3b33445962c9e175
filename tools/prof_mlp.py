"""Time the decoder FFN (Linear 256->1024, GELU, Linear 1024->256; M packed rows) forward+backward with the GELU in the GEMM
epilogues vs as stand-alone passes, fp32 and bf16 operands.  usage (GPU box): python tools/prof_mlp.py [M]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from unidet3d_amd import dense, precision  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16800
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).to(dev).requires_grad_()
w1 = (torch.randn(1024, 256, generator=g) * 0.05).to(dev).requires_grad_(); b1 = torch.zeros(1024, device=dev, requires_grad=True)
w2 = (torch.randn(256, 1024, generator=g) * 0.05).to(dev).requires_grad_(); b2 = torch.zeros(256, device=dev, requires_grad=True)
go = torch.randn(M, 256, generator=g).to(dev)
for mode in ('fp32', 'bf16'):
    for fuse in (True, False):
        dense.FUSE_GELU = fuse
        with precision.operands(mode):
            for it in range(3):
                z = dense.mlp(x, w1, b1, w2, b2, 'gelu'); z.backward(go)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(20):
                z = dense.mlp(x, w1, b1, w2, b2, 'gelu'); z.backward(go)
            torch.cuda.synchronize()
            print(f'{mode} fuse_gelu={fuse}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per FFN fwd+bwd (M={M})')

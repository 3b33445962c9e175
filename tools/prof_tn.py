"""Time the weight-gradient GEMM (dW[N,K] = dY[M,N]^T X[M,K], u3d_gemm_tn + its fixed-order reduce) at the decoder's shapes and
check it against torch in fp64.  usage (GPU box): python tools/prof_tn.py [M]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from unidet3d_amd import dense  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16800
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
for N, K in ((256, 256), (768, 256), (1024, 256), (256, 1024), (256, 32), (20, 256), (8, 256), (64, 64), (132, 68)):
    dy = torch.randn(M, N, generator=g).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    for bf in (False, True):
        dw, db = dense._weight_grad(dy, x, True, bf)
        ref = dy.double().t() @ x.double()
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        eb = float((db.double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(50):
            dense._weight_grad(dy, x, True, bf)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        print(f'tn M={M} N={N} K={K} {"bf16" if bf else "fp32"}: {us:7.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TF/s  err {err:.1e} bias {eb:.1e}')

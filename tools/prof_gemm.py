"""Per-shape timing of the decoder GEMMs: own kernels vs torch (hipBLASLt)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd import _lib as L
from unidet3d_amd.dense import _gemm_nt
dev = torch.device('cuda:0')
M = 16000
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (N, K) in [(768, 256), (256, 256), (1024, 256), (256, 1024), (256, 32)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    t1 = bench(lambda: _gemm_nt(a, w, b)); t2 = bench(lambda: torch.nn.functional.linear(a, w, b))
    fl = 2.0 * M * N * K
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev); ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, N, K), dev)
    t3 = bench(lambda: L.call('u3d_gemm_tn', L.ptr(dy), L.ptr(a), L.ptr(dw), L.ptr(db), M, N, K, L.ptr(ws), 0.0, L.stream()))
    t4 = bench(lambda: dy.t() @ a)
    print(f'N={N:5d} K={K:5d}: nt own {t1*1e6:7.1f} us {fl/t1/1e12:6.1f} TF/s | torch {t2*1e6:7.1f} us {fl/t2/1e12:6.1f} TF/s || tn own {t3*1e6:7.1f} us {fl/t3/1e12:6.1f} | torch {t4*1e6:7.1f} us {fl/t4/1e12:6.1f}')

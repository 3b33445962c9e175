"""Per-shape timing of the decoder GEMMs (NT: Linear forward / input gradient, TN: weight gradient) in both fp32 math modes,
next to torch (hipBLASLt):  python tools/prof_gemm.py [M ...]      env U3D_NT_TILE=1|2|3 forces the NT tile."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd import _lib as L
from unidet3d_amd import precision as P
from unidet3d_amd.dense import _gemm_nt
dev = torch.device('cuda:0')
SHAPES = [(256, 256), (768, 256), (1024, 256), (256, 1024), (256, 32)]
if os.environ.get('PROF_SHAPES'):            # e.g. PROF_SHAPES=768,256;256,1024
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['PROF_SHAPES'].split(';')]


def bench(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for M in [int(a) for a in sys.argv[1:]] or [41000]:
    for (N, K) in SHAPES:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev); ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, N, K), dev)
        fl = 2.0 * M * N * K
        line = f'M={M:6d} N={N:5d} K={K:5d} ({fl / 1e9:5.1f} GF):'
        for mode in ('mfma', 'bf16x3'):
            with P.fp32_math(mode):
                t1 = bench(lambda: _gemm_nt(a, w, b))
                t3 = bench(lambda: L.call('u3d_gemm_tn', L.ptr(dy), L.ptr(a), L.ptr(dw), L.ptr(db), M, N, K, L.ptr(ws), 0.0, L.stream()))
            line += f' | {mode}: nt {t1 * 1e6:6.1f} us {fl / t1 / 1e12:6.1f} TF/s, tn {t3 * 1e6:6.1f} us {fl / t3 / 1e12:6.1f}'
        t2 = bench(lambda: torch.nn.functional.linear(a, w, b)); t4 = bench(lambda: dy.t() @ a)
        print(line + f' | torch: nt {t2 * 1e6:6.1f} us, tn {t4 * 1e6:6.1f} us', flush=True)

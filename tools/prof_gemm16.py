"""Per-shape timing of the decoder GEMMs under bf16 operands: fp32 tensors rounded in flight (gemm.hip, the round-5 data flow) against
bf16 tensors in HBM (gemm_b16.hip, include/u3d.h K14b).   python tools/prof_gemm16.py [M ...]      env U3D_NT16_TILE=1|2|3"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd import _lib as L
from unidet3d_amd import precision as P
from unidet3d_amd import dense16 as D16
from unidet3d_amd.dense import _gemm_nt
dev = torch.device('cuda:0')
SHAPES = [(256, 256), (768, 256), (1024, 256), (256, 1024), (256, 32)]


def bench(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for M in [int(a) for a in sys.argv[1:]] or [24600]:
    for (N, K) in SHAPES:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        a16, dy16 = a.bfloat16(), dy.bfloat16()
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev); ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, N, K), dev)
        fl = 2.0 * M * N * K
        us = lambda t: f'{t * 1e6:6.1f}'
        with P.operands('bf16'):
            t_old = bench(lambda: _gemm_nt(a, w, b, True))
        t_ff = bench(lambda: D16.gemm_nt(a, w, b))
        t_hf = bench(lambda: D16.gemm_nt(a16, w, b))
        t_fh = bench(lambda: D16.gemm_nt(a, w, b, out_bf16=True))
        t_hh = bench(lambda: D16.gemm_nt(a16, w, b, out_bf16=True))
        w16 = w.bfloat16(); t_torch = bench(lambda: torch.nn.functional.linear(a16, w16))
        t_tn_old = bench(lambda: L.call('u3d_gemm_tn_bf16', L.ptr(dy), L.ptr(a), L.ptr(dw), L.ptr(db), M, N, K, L.ptr(ws), 0.0, L.stream()))
        t_tn = [bench(lambda: D16.gemm_tn(p, q, True)) for p, q in ((dy, a), (dy16, a), (dy, a16), (dy16, a16))]
        print(f'M={M:6d} N={N:5d} K={K:5d} ({fl / 1e9:5.1f} GF) us | nt: fp32-tensor kernel {us(t_old)} | b16 kernel A/C = f/f {us(t_ff)} h/f {us(t_hf)} f/h {us(t_fh)} '
              f'h/h {us(t_hh)} ({fl / t_hh / 1e12:5.0f} TF/s) torch bf16 {us(t_torch)} | tn: old {us(t_tn_old)} | b16 f/f {us(t_tn[0])} h/f {us(t_tn[1])} f/h {us(t_tn[2])} h/h {us(t_tn[3])} '
              f'({fl / t_tn[3] / 1e12:5.0f} TF/s)', flush=True)
    n = M * 1024
    h = torch.randn(M, 1024, device=dev); hb = h.bfloat16(); o = torch.empty_like(h); ob = torch.empty_like(hb)
    t0 = bench(lambda: L.call('u3d_gelu_fwd', L.ptr(h), L.ptr(o), n, L.stream()))
    t1 = bench(lambda: L.call('u3d_gelu_fwd_b16', L.ptr(hb), L.ptr(ob), n, L.stream()))
    t2 = bench(lambda: L.call('u3d_gelu_bwd', L.ptr(h), L.ptr(h), L.ptr(o), n, L.stream()))
    t3 = bench(lambda: L.call('u3d_gelu_bwd_b16', L.ptr(hb), L.ptr(hb), L.ptr(ob), n, L.stream()))
    print(f'gelu [{M} x 1024] us: fwd fp32 {t0 * 1e6:.1f} bf16 {t1 * 1e6:.1f} | bwd fp32 {t2 * 1e6:.1f} bf16 {t3 * 1e6:.1f}')

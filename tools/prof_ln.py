"""LayerNorm forward / backward kernel timing at the decoder's shapes (C ABI calls, no autograd): python tools/prof_ln.py [M ...]
env U3D_LN_BLOCKS: grid cap of the backward kernel"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd import _lib as L
dev = torch.device('cuda:0')


def bench(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in [int(a) for a in sys.argv[1:]] or [16937, 24600, 118559]:
    C = 256
    x = torch.randn(M, C, device=dev); r = torch.randn(M, C, device=dev); w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    s = torch.empty_like(x); y = torch.empty_like(x); st = torch.empty(M, 2, device=dev); go = torch.randn(M, C, device=dev)
    dx = torch.empty_like(x); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    ws = L.scratch(L.lib().u3d_layer_norm_ws_bytes(M, C), dev)
    fwd = lambda: L.call('u3d_layer_norm_fwd', L.ptr(x), L.ptr(r), L.ptr(w), L.ptr(b), M, C, 1e-5, L.ptr(s), L.ptr(y), L.ptr(st), L.stream())
    bwd = lambda: L.call('u3d_layer_norm_bwd', L.ptr(s), L.ptr(go), L.ptr(w), L.ptr(st), M, C, L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(ws), L.stream())
    t_f = bench(fwd); t_b = bench(bwd)
    print(f'M={M:7d} C={C}: forward {t_f:6.1f} us ({M * C * 16 / t_f / 1e6:5.2f} TB/s of 16 B/elt), backward + reduce {t_b:6.1f} us ({M * C * 12 / t_b / 1e6:5.2f} TB/s of 12 B/elt)')

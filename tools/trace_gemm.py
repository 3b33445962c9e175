"""Phase timing of gemm_nt_x3_k from s_memtime stamps (a -DU3D_NTX_TRACE build of gemm.hip, tools/build_variant.sh):
U3D_LIB_PATH=tools/bin/libu3d_nttrace.so python tools/trace_gemm.py [M N K]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd import _lib as L
dev = torch.device('cuda:0')
M, N, K = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (41000, 768, 256)
tile = {1: (128, 128), 2: (128, 64), 3: (64, 64)}[int(os.environ.get('U3D_NT_TILE', '1'))]
wgs = -(-M // tile[0]) * -(-N // tile[1])
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
tr = torch.zeros(wgs * 12, dtype=torch.int64, device=dev)
for _ in range(3):
    L.call('u3d_linear_act', L.ptr(a), L.ptr(w), L.ptr(b), 0, L.ptr(tr), L.ptr(y), M, N, K, 0.0, L.stream())
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(wgs, 12)
t0 = t[:, 0].min()
names = ['prologue (loads, split, store, barrier)', 'steps 0-2', 'k3: frag reads + MFMAs + split', 'k3: issue loads', 'k3: wait barrier 1', 'k3: LDS stores', 'k3: wait barrier 2',
         'steps 4..', 'epilogue']
d = np.diff(t[:, :10], axis=1).astype(np.float64)
print(f'M={M} N={N} K={K} tile {tile}: {wgs} workgroups; s_memtime ticks (shader-clock scale; compare phases, not absolutes); kernel span {(t[:, 9].max() - t0) / 100:.1f} us')
for i, n in enumerate(names):
    print(f'  {n:42s} median {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}')
print('  whole workgroup median', np.median(t[:, 9] - t[:, 0]), ' start spread: first', 0, 'last', t[:, 0].max() - t0)
hw = t[:, 10]
cu = ((hw >> 28) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)      # (xcc, se, sh, cu)
print('  distinct CUs', len(np.unique(cu)))
order = np.argsort(t[:, 0])
first = {}
for b in order[:512 if wgs > 512 else wgs]:
    first.setdefault(int(cu[b]), []).append(int(b))
pairs = [v for v in first.values() if len(v) >= 2][:12]
print('  co-resident workgroup ids (first wave):', pairs)
print('  xcc of blocks 0..15:', [int((hw[b] >> 28) & 0xf) for b in range(16)])

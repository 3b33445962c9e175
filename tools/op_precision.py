"""Measured error of each decoder op (product kernels vs the same op in torch fp32 on the GPU), both against fp64.
usage (GPU box): python tools/op_precision.py"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from unidet3d_amd import dense  # noqa: E402
from unidet3d_amd.encoder import attention_varlen  # noqa: E402

DEV = 'cuda:0'


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def report(name, prod, base, ref):
    print(f'{name:42s} product {max(rel(p, r) for p, r in zip(prod, ref)):9.2e}   torch-fp32 {max(rel(p, r) for p, r in zip(base, ref)):9.2e}'
          f'   per-tensor product {[f"{rel(p, r):.1e}" for p, r in zip(prod, ref)]}')


g = torch.Generator().manual_seed(0)
for M in (600, 16800):
    x = torch.randn(M, 256, generator=g); w = torch.randn(1024, 256, generator=g) * 0.06; b = torch.randn(1024, generator=g) * 0.1
    w2 = torch.randn(256, 1024, generator=g) * 0.03; b2 = torch.randn(256, generator=g) * 0.1
    go = torch.randn(M, 256, generator=g); go1 = torch.randn(M, 1024, generator=g)
    res = torch.randn(M, 256, generator=g); lw = torch.rand(256, generator=g) + 0.5; lb = torch.randn(256, generator=g) * 0.1

    def run_linear(fn, dt, dev):
        t = [v.clone().to(dt).to(dev).requires_grad_() for v in (x, w, b)]
        y = fn(*t); y.backward(go1.to(dt).to(dev))
        return [y] + [v.grad for v in t]
    report(f'linear 256->1024 M={M}', run_linear(dense.linear, torch.float32, DEV), run_linear(F.linear, torch.float32, DEV), run_linear(F.linear, torch.float64, 'cpu'))

    for act in ('relu', 'gelu'):
        def run_mlp(prod, dt, dev):
            t = [v.clone().to(dt).to(dev).requires_grad_() for v in (x, w, b, w2, b2)]
            if prod:
                z = dense.mlp(*t, act)
            else:
                h = F.linear(t[0], t[1], t[2]); z = F.linear(torch.relu(h) if act == 'relu' else F.gelu(h), t[3], t[4])
            z.backward(go.to(dt).to(dev))
            return [z] + [v.grad for v in t]
        report(f'mlp {act} M={M}', run_mlp(True, torch.float32, DEV), run_mlp(False, torch.float32, DEV), run_mlp(False, torch.float64, 'cpu'))

    def run_ln(prod, dt, dev):
        t = [v.clone().to(dt).to(dev).requires_grad_() for v in (x, res, lw, lb)]
        y = dense.layer_norm(t[0], t[2], t[3], 1e-5, t[1]) if prod else F.layer_norm(t[0] + t[1], (256,), t[2], t[3], 1e-5)
        y.backward(go.to(dt).to(dev))
        return [y] + [v.grad for v in t]
    report(f'layer_norm(x + res) M={M}', run_ln(True, torch.float32, DEV), run_ln(False, torch.float32, DEV), run_ln(False, torch.float64, 'cpu'))

for lens in ([600], [2100, 1900, 2000]):
    n = sum(lens); H, hd = 8, 32
    qkv = torch.randn(n, 768, generator=g); go = torch.randn(n, 256, generator=g)

    def run_attn(prod, dt, dev):
        t = qkv.clone().to(dt).to(dev).requires_grad_()
        if prod:
            cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)
            o = attention_varlen(t, cu, max(lens), H)
        else:
            outs, s = [], 0
            for ln in lens:
                q, k, v = t[s:s + ln].chunk(3, -1); s += ln
                q, k, v = [u.view(ln, H, hd).transpose(0, 1) for u in (q, k, v)]
                outs.append((torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), -1) @ v).transpose(0, 1).reshape(ln, 256))
            o = torch.cat(outs)
        o.backward(go.to(dt).to(dev))
        return [o, t.grad]
    report(f'attention lens={lens}', run_attn(True, torch.float32, DEV), run_attn(False, torch.float32, DEV), run_attn(False, torch.float64, 'cpu'))

"""nn.Linear through the gfx950 fp32 MFMA GEMM kernels (include/u3d.h K14): forward, input gradient and
weight gradient of the decoder's Linear layers (unidet3d/encoder.py:19-21,55-61,138-140,153-155,163)."""
from __future__ import annotations

import torch

from . import _lib as L

_PROFILE_FLOPS = False


def set_profile_flops(on: bool):
    global _PROFILE_FLOPS
    _PROFILE_FLOPS = bool(on)


def _gemm_nt(a, w, bias):
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if M:
        L.call('u3d_gemm_nt', L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(c), M, N, K, 2.0 * M * N * K if _PROFILE_FLOPS else 0.0,
               L.stream())
    return c


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _gemm_nt(x, weight.contiguous(), bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if N % 16 == 0:
                wt = torch.empty(K, N, dtype=torch.float32, device=weight.device)
                L.call('u3d_transpose', L.ptr(weight.contiguous()), L.ptr(wt), N, K, L.stream())
                dx = _gemm_nt(dy, wt, None)                     # dX[M,K] = dY[M,N] . (W^T)[K,N]^T
            else:                                               # tiny heads (N = 19, 8): pad N to 16 with zero columns
                Np = (N + 15) // 16 * 16
                wt = torch.zeros(K, Np, dtype=torch.float32, device=weight.device)
                wt[:, :N] = weight.t()
                dyp = torch.zeros(M, Np, dtype=torch.float32, device=dy.device)
                dyp[:, :N] = dy
                dx = _gemm_nt(dyp, wt, None)
        if ctx.needs_input_grad[1]:
            dw = torch.empty(N, K, dtype=torch.float32, device=weight.device)
            if M and N % 4 == 0:
                ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, N, K), weight.device)
                if ctx.has_bias and ctx.needs_input_grad[2]:      # the bias gradient (column sums of dy) rides along
                    db = torch.empty(N, dtype=torch.float32, device=weight.device)
                L.call('u3d_gemm_tn', L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), M, N, K, L.ptr(ws),
                       2.0 * M * N * K if _PROFILE_FLOPS else 0.0, L.stream())
            elif M:
                Np = (N + 3) // 4 * 4
                dyp = torch.zeros(M, Np, dtype=torch.float32, device=dy.device)
                dyp[:, :N] = dy
                dwp = torch.empty(Np, K, dtype=torch.float32, device=weight.device)
                ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, Np, K), weight.device)
                dbp = torch.empty(Np, dtype=torch.float32, device=weight.device) if ctx.has_bias and ctx.needs_input_grad[2] else None
                L.call('u3d_gemm_tn', L.ptr(dyp), L.ptr(x), L.ptr(dwp), L.ptr(dbp), M, Np, K, L.ptr(ws), 0.0, L.stream())
                dw = dwp[:N].contiguous()
                if dbp is not None:
                    db = dbp[:N].contiguous()
            else:
                dw.zero_()
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """y = x W^T + b for 2-D x [M, K] (K % 16 == 0)."""
    return _LinearFn.apply(x, weight, bias)


class _LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res) over the last dimension (include/u3d.h K15); the gradient of x and res is the same tensor."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        x = x.contiguous()
        M, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        s = x
        if res is not None:
            res = res.contiguous()
            s = torch.empty_like(x)
        if M:
            L.call('u3d_layer_norm_fwd', L.ptr(x), L.ptr(res), L.ptr(weight), L.ptr(bias), M, C, float(eps),
                   L.ptr(s) if res is not None else None, L.ptr(y), L.ptr(stats), L.stream())
        ctx.save_for_backward(s, weight, stats)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        s, weight, stats = ctx.saved_tensors
        dy = dy.contiguous()
        M, C = s.shape
        dx = torch.empty_like(s)
        dg = torch.empty(C, dtype=torch.float32, device=s.device)
        db = torch.empty(C, dtype=torch.float32, device=s.device)
        if M:
            ws = L.scratch(L.lib().u3d_layer_norm_ws_bytes(M, C), s.device)
            L.call('u3d_layer_norm_bwd', L.ptr(s), L.ptr(dy), L.ptr(weight), L.ptr(stats), M, C, L.ptr(dx), L.ptr(dg), L.ptr(db),
                   L.ptr(ws), L.stream())
        else:
            dg.zero_(); db.zero_()
        return dx, (dx if ctx.has_res else None), dg, db, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, res: torch.Tensor = None) -> torch.Tensor:
    """LayerNorm(x + res) for 2-D x [M, C] (C % 4 == 0, C <= 1024)."""
    return _LayerNormFn.apply(x, res, weight, bias, eps)


class LayerNorm(torch.nn.LayerNorm):
    """``nn.LayerNorm(d_model)`` of the reference (same parameters / state_dict keys) on the HIP kernels, optionally fused with the
    residual add in front of it."""

    def forward(self, x, res=None):
        return layer_norm(x, self.weight, self.bias, self.eps, res)

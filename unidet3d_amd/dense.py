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
                L.call('u3d_gemm_tn', L.ptr(dyp), L.ptr(x), L.ptr(dwp), None, M, Np, K, L.ptr(ws), 0.0, L.stream())
                dw = dwp[:N].contiguous()
            else:
                dw.zero_()
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """y = x W^T + b for 2-D x [M, K] (K % 16 == 0)."""
    return _LinearFn.apply(x, weight, bias)

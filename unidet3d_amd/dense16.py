"""The decoder's Linear / MLP / LayerNorm ops with bf16 ACTIVATIONS in HBM (include/u3d.h K14b, csrc/gemm_b16.hip).

BASELINE configs[2] is the reference's ``--amp`` run (tools/train.py:86-99): autocast hands every ``nn.Linear`` /
``nn.MultiheadAttention`` of unidet3d/encoder.py:19-21,55-61,138-163 a 16-bit input and gets a 16-bit output back, while LayerNorm
results and the residual stream stay fp32.  ``precision.bf16_act()`` turns the same data flow on here (it implies bf16 MFMA
operands, ``precision.bf16()``):

* a LayerNorm writes its fp32 result AND a bf16 copy of it (``attach_b16`` -- an attribute of the fp32 tensor, like the batch-norm
  shadows of sparse.py); the Linear that follows streams the copy.  The LayerNorm backward does the same for the gradient it hands
  to the layer in front of it;
* the hidden tensors of an MLP (FFN pre-activation / activation, 1024 wide; ReLU hidden of ``input_proj`` / ``outs_cls``) and their
  gradients exist ONLY in bf16: written by a GEMM epilogue, read by the GELU pass and by the next GEMM;
* the packed q / k / v projection, the attention output and their gradients are bf16 tensors as well (the attention kernels read and
  write them as they are: include/u3d.h u3d_attn_varlen_*_b16);
* results that feed a LayerNorm, the residual stream or the criterion stay fp32.

Products are the ones ``dense.py`` forms under ``precision.bf16()`` (bf16 operands rounded to nearest even, fp32 accumulation): a
bf16 copy holds exactly the rounding the fp32-tensor kernels apply in flight.  What differs is where a value is rounded ONCE MORE: the
GELU and its derivative see the rounded pre-activation, and the hidden gradient is rounded before the derivative multiplies it.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import account
from . import dense as D

A16, B16, C16 = 1, 2, 4          # include/u3d.h U3D_A_BF16 / U3D_B_BF16 / U3D_C_BF16
EPI_BIAS, EPI_RELU, EPI_RELU_MASK, EPI_ADD = 0, 1, 3, 5

STATS = {'hit': 0, 'miss': 0}


def attach_b16(t: torch.Tensor, copy: torch.Tensor):
    """``copy`` = ``t`` rounded to bf16 (same shape): remembered on the tensor object, valid while ``t`` is not modified in place."""
    t._u3d_b16 = (copy, t._version)


def b16_of(t: torch.Tensor):
    e = getattr(t, '_u3d_b16', None)
    if e is not None and e[1] == t._version and e[0].shape == t.shape and e[0].device == t.device:
        STATS['hit'] += 1
        return e[0]
    STATS['miss'] += 1
    return None


def _operand(t: torch.Tensor):
    """(tensor to stream, is-bf16): a bf16 tensor itself, the bf16 copy of an fp32 tensor when it has one, else the fp32 tensor"""
    if t.dtype == torch.bfloat16:
        return t, True
    c = b16_of(t)
    return (c, True) if c is not None else (t, False)


def _book(M, N, K, bytes_):
    if not D._PROFILE_FLOPS:
        return 0.0
    account.add('gemm', 2.0 * M * N * K, float(bytes_))
    return 2.0 * M * N * K


def gemm_nt(a, w, bias=None, epi=EPI_BIAS, aux=None, out_bf16=False):
    """epi(a [M,K] . w [N,K]^T): ``a`` fp32 or bf16, ``w`` / ``bias`` fp32; result bf16 when ``out_bf16`` else fp32."""
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    if M:
        flags = (A16 if a.dtype == torch.bfloat16 else 0) | (C16 if out_bf16 else 0)
        fl = _book(M, N, K, M * K * a.element_size() + N * K * 4 + M * N * c.element_size() * (2 if aux is not None else 1))
        L.call('u3d_gemm_nt_b16', L.ptr(a), L.ptr(w), L.ptr(bias), epi, L.ptr(aux), L.ptr(c), flags, M, N, K, fl, L.stream())
    return c


def gemm_tn(dy, x, want_bias):
    """(dy^T x [N,K] fp32, column sums of dy [N] or None); either operand fp32 or bf16"""
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    qa, qb = (8 if dy.dtype == torch.bfloat16 else 4), (8 if x.dtype == torch.bfloat16 else 4)
    if N % qa:                                          # tiny heads (N = 19): pad the columns of dy with zeros
        Np = (N + qa - 1) // qa * qa
        dw, db = gemm_tn(torch.nn.functional.pad(dy, (0, Np - N)), x, want_bias)
        return dw[:N].contiguous(), (db[:N].contiguous() if db is not None else None)
    if K % qb:
        raise L.U3DError(f'gemm_tn_b16: K={K} must be a multiple of {qb}')
    dw = torch.empty(N, K, dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    if M:
        ws = L.scratch(L.lib().u3d_gemm_tn_b16_ws_bytes(M, N, K), dev)
        flags = (A16 if dy.dtype == torch.bfloat16 else 0) | (B16 if x.dtype == torch.bfloat16 else 0)
        fl = _book(M, N, K, M * N * dy.element_size() + M * K * x.element_size() + N * K * 4)
        L.call('u3d_gemm_tn_b16', L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), flags, M, N, K, L.ptr(ws), fl, L.stream())
    else:
        dw.zero_()
        if db is not None:
            db.zero_()
    return dw, db


def _weight_grad(dy, x, want_bias, weight, bias):
    """``gemm_tn`` on the weight-gradient side stream when the overlap of sparse.set_wgrad_overlap(2) applies (dense._weight_grad_overlapped)"""
    from . import sparse
    ok = sparse._WGRAD_OVERLAP == 2 and D._OVERLAP_TN and dy.is_cuda and sparse.async_dw_ok(weight, bias)
    if not ok:
        return gemm_tn(dy, x, want_bias)
    dev = dy.device
    main = torch.cuda.current_stream(dev)
    side = sparse._side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw, db = gemm_tn(dy, x, want_bias)
    dy.record_stream(side)
    x.record_stream(side)
    for t in (dw, db):
        if t is not None:
            t.record_stream(main)
    sparse._queue_join(dev)
    return dw, db


def _transposed(weight, wt):
    """[K, N] copy of ``weight`` [N, K] with the reduction dim N zero-padded to a multiple of 32 (tiny heads); ``wt``: the copy of
    dense.transposed_weights() when there is one"""
    N, K = weight.shape
    if N % 32 == 0:
        if wt is None:
            wt = torch.empty(K, N, dtype=torch.float32, device=weight.device)
            L.call('u3d_transpose', L.ptr(weight.contiguous()), L.ptr(wt), N, K, L.stream())
        return wt, N
    Np = (N + 31) // 32 * 32
    wt = torch.zeros(K, Np, dtype=torch.float32, device=weight.device)
    wt[:, :N] = weight.t()
    return wt, Np


def _pad_cols(t, n):
    return t if t.shape[1] == n else torch.nn.functional.pad(t, (0, n - t.shape[1]))


class _Linear16Fn(torch.autograd.Function):
    """y = x W^T + b; x and (in backward) dy are streamed as bf16 whenever they are bf16 tensors or carry a bf16 copy.  The result is
    fp32, or -- ``out_bf16``: the packed q / k / v projection, whose only consumer is the attention kernel -- a bf16 tensor (its
    gradient then arrives as one); a bf16 INPUT (the attention output in front of ``out_proj``) gets a bf16 gradient back."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_bf16):
        x = x.contiguous()
        xa, _ = _operand(x)
        w = weight.contiguous()
        ctx.save_for_backward(xa, w)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.wt = D._wt_of(weight)
        ctx.dx_bf16 = x.dtype == torch.bfloat16
        return gemm_nt(xa, w, bias, out_bf16=bool(out_bf16) and w.shape[0] % 2 == 0)

    @staticmethod
    def backward(ctx, dy):
        xa, weight = ctx.saved_tensors
        dy = dy.contiguous()
        da, _ = _operand(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dw, db = _weight_grad(da, xa, ctx.has_bias and ctx.needs_input_grad[2], weight, ctx.bias_ref)
        if ctx.needs_input_grad[0]:
            wt, Np = _transposed(weight, ctx.wt)
            dx = gemm_nt(_pad_cols(da, Np), wt, out_bf16=ctx.dx_bf16)
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db, None


class _MLP16Fn(torch.autograd.Function):
    """z = act(x W1^T + b1) W2^T + b2 with the hidden tensors (and their gradients) in bf16 only; z fp32."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act):
        x = x.contiguous()
        xa, _ = _operand(x)
        M = x.shape[0]
        w1c, w2c = w1.contiguous(), w2.contiguous()
        h = None
        if act == D.ACT_GELU:
            h = gemm_nt(xa, w1c, b1, EPI_BIAS, out_bf16=True)
            a = torch.empty_like(h)
            if M:
                L.call('u3d_gelu_fwd_b16', L.ptr(h), L.ptr(a), h.numel(), L.stream())
        else:
            a = gemm_nt(xa, w1c, b1, EPI_RELU, out_bf16=True)
        z = gemm_nt(a, w2c, b2)
        ctx.save_for_backward(xa, w1c, w2c, a, h)
        ctx.act, ctx.bias = act, (b1 is not None, b2 is not None)
        ctx.bias_refs = (b1, b2)
        ctx.wt1, ctx.wt2 = D._wt_of(w1c), D._wt_of(w2c)
        return z

    @staticmethod
    def backward(ctx, dz):
        xa, w1, w2, a, h = ctx.saved_tensors
        dz = dz.contiguous()
        dza, _ = _operand(dz)
        need = ctx.needs_input_grad
        dw2, db2 = _weight_grad(dza, a, ctx.bias[1] and need[4], w2, ctx.bias_refs[1]) if need[3] else (None, None)
        wt2, Np = _transposed(w2, ctx.wt2)
        dzp = _pad_cols(dza, Np)
        if ctx.act == D.ACT_GELU:
            da = gemm_nt(dzp, wt2, out_bf16=True)
            dh = torch.empty_like(da)
            if da.numel():
                L.call('u3d_gelu_bwd_b16', L.ptr(da), L.ptr(h), L.ptr(dh), da.numel(), L.stream())
        else:
            dh = gemm_nt(dzp, wt2, None, EPI_RELU_MASK, aux=a, out_bf16=True)
        dw1, db1 = _weight_grad(dh, xa, ctx.bias[0] and need[2], w1, ctx.bias_refs[0]) if need[1] else (None, None)
        dx = None
        if need[0]:
            wt1, _ = _transposed(w1, ctx.wt1)
            dx = gemm_nt(dh, wt1)
        return dx, dw1, db1, dw2, db2, None


class _LayerNorm16Fn(torch.autograd.Function):
    """dense._LayerNormFn + the bf16 copies: (y, ..., y16) forward -- y16 is attached to every y output by ``layer_norm`` below --, dx
    with its copy attached in backward."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps, n_out=1):
        x = x.contiguous()
        M, C = x.shape
        y = torch.empty_like(x)
        y16 = torch.empty(M, C, dtype=torch.bfloat16, device=x.device)
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        s = x
        if res is not None:
            res = res.contiguous()
            s = torch.empty_like(x)
        if M:
            L.call('u3d_layer_norm_fwd_b16', L.ptr(x), L.ptr(res), L.ptr(weight), L.ptr(bias), M, C, float(eps),
                   L.ptr(s) if res is not None else None, L.ptr(y), L.ptr(y16), L.ptr(stats), L.stream())
        ctx.save_for_backward(s, weight, stats)
        ctx.has_res = res is not None
        ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)
        ys = D._aliases(y, n_out)
        return (ys if n_out > 1 else (ys,)) + (y16,)

    @staticmethod
    def backward(ctx, *dys):
        s, weight, stats = ctx.saved_tensors
        dy, dy2, dy3 = D._grads_in(dys[:-1])
        if dy is None:
            return (None,) * 6
        M, C = s.shape
        dx = torch.empty_like(s)
        dx16 = torch.empty(M, C, dtype=torch.bfloat16, device=s.device)
        dg = torch.empty(C, dtype=torch.float32, device=s.device)
        db = torch.empty(C, dtype=torch.float32, device=s.device)
        if M:
            ws = L.scratch(L.lib().u3d_layer_norm_ws_bytes(M, C), s.device)
            L.call('u3d_layer_norm_bwd_sum', L.ptr(s), L.ptr(dy), L.ptr(dy2), L.ptr(dy3), L.ptr(weight), L.ptr(stats), M, C, L.ptr(dx), L.ptr(dx16),
                   L.ptr(dg), L.ptr(db), L.ptr(ws), L.stream())
        else:
            dg.zero_(); db.zero_()
        attach_b16(dx, dx16)
        return dx, (dx if ctx.has_res else None), dg, db, None, None


class _LNLinear16Fn(torch.autograd.Function):
    """(nq, nq16, y) = (LayerNorm(x), its bf16 copy, nq W^T + b): dense._LNLinearFn with the Linear and its weight gradient streaming
    nq16 (the head: out_norm -> out_bboxes.linear, unidet3d/encoder.py:187-196)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, weight, bias):
        x = x.contiguous()
        M, C = x.shape
        dev = x.device
        nq = torch.empty_like(x)
        nq16 = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        stats = torch.empty(M, 2, dtype=torch.float32, device=dev)
        w = weight.contiguous()
        if M:
            L.call('u3d_layer_norm_fwd_b16', L.ptr(x), None, L.ptr(gamma), L.ptr(beta), M, C, float(eps), None, L.ptr(nq), L.ptr(nq16),
                   L.ptr(stats), L.stream())
        y = gemm_nt(nq16, w, bias)
        ctx.save_for_backward(x, gamma, stats, nq16, w)
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(nq16)
        return nq, nq16, y

    @staticmethod
    def backward(ctx, dnq, _unused, dy):
        x, gamma, stats, nq16, w = ctx.saved_tensors
        M, C = x.shape
        dev = x.device
        dw = db = None
        if dy is not None:
            dy = dy.contiguous()
            dw, db = gemm_tn(dy, nq16, ctx.has_bias)
            wt, Np = _transposed(w, None)
            dyp = _pad_cols(dy, Np)
            if dnq is None:
                dtot = gemm_nt(dyp, wt)
            else:                                        # dtot = dy W + dnq in one GEMM
                dtot = gemm_nt(dyp, wt, None, EPI_ADD, aux=dnq.contiguous())
        else:
            dtot = dnq.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        if M:
            ws = L.scratch(L.lib().u3d_layer_norm_ws_bytes(M, C), dev)
            L.call('u3d_layer_norm_bwd', L.ptr(x), L.ptr(dtot), L.ptr(gamma), L.ptr(stats), M, C, L.ptr(dx), L.ptr(dg), L.ptr(dbeta),
                   L.ptr(ws), L.stream())
        else:
            dg.zero_(); dbeta.zero_()
        return dx, dg, dbeta, None, dw, db


def _ok(*dims):
    return all(d % 32 == 0 for d in dims)


def linear(x, weight, bias=None, out_bf16=False):
    return _Linear16Fn.apply(x, weight, bias, out_bf16)


def mlp(x, w1, b1, w2, b2, act):
    return _MLP16Fn.apply(x, w1, b1, w2, b2, act)


def layer_norm(x, weight, bias, eps, res=None, n_out=1):
    *ys, y16 = _LayerNorm16Fn.apply(x, res, weight, bias, eps, n_out)
    for y in ys:
        attach_b16(y, y16)
    return ys[0] if n_out == 1 else tuple(ys)


def ln_linear(x, gamma, beta, eps, weight, bias):
    nq, nq16, y = _LNLinear16Fn.apply(x, gamma, beta, eps, weight, bias)
    attach_b16(nq, nq16)
    return nq, y

"""nn.Linear through the gfx950 fp32 MFMA GEMM kernels (include/u3d.h K14): forward, input gradient and
weight gradient of the decoder's Linear layers (unidet3d/encoder.py:19-21,55-61,138-140,153-155,163)."""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import precision as P
from . import account

_PROFILE_FLOPS = False


def set_profile_flops(on: bool):
    global _PROFILE_FLOPS
    _PROFILE_FLOPS = bool(on)


def _pad_k(t, mult):
    """zero-pad the last (reduction) dim of a 2-D tensor to a multiple of ``mult``"""
    k = t.shape[1]
    return t if k % mult == 0 else torch.nn.functional.pad(t, (0, mult - k % mult))


def _gemm_nt(a, w, bias, bf=False, planes=None):
    """a [M,K] . w[N,K]^T (+ bias); ``bf``: bf16 MFMA operands (K is zero-padded to a multiple of 32 when needed);
    ``planes``: the pre-split planes of ``w`` (three-plane products only)."""
    M = a.shape[0]
    N = w.shape[0]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if M:
        if bf:
            a, w = _pad_k(a, 32), _pad_k(w, 32)
            L.call('u3d_linear_act', L.ptr(a), L.ptr(w), L.ptr(bias), P.BF16_FLAG, None, L.ptr(c), M, N, a.shape[1],
                   _flops(M, N, a.shape[1]), L.stream())
        else:
            K = a.shape[1]
            _use_planes(w, planes)
            L.call('u3d_gemm_nt', L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(c), M, N, K, _flops(M, N, K), L.stream())
    return c


ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


# ---- transposed weight copies of a whole forward pass in one launch ----------------------------------------------------------------
# dX = dY . W runs as an NT product over W^T, so every Linear's backward needs a [K, N] copy of its [N, K] weight: ~31 transpose
# launches of ~4 us per training step.  ``transposed_weights(module)`` makes them all with ONE u3d_transpose_batch launch when the
# forward pass starts; the Linear / MLP functions pick their copy up in FORWARD (while the context is active) and keep it for their
# backward, so a copy can never outlive the weights it was made from (weights are not modified between a forward and its backward).
# Outside the context every op falls back to its own u3d_transpose launch.
_WT_ACTIVE = None          # data_ptr of a weight -> its [K, N] copy, while a transposed_weights() context is open


class transposed_weights:
    def __init__(self, module: torch.nn.Module):
        self.module = module

    def __enter__(self):
        global _WT_ACTIVE
        self.prev = _WT_ACTIVE
        ws = []
        if torch.is_grad_enabled():
            seen = set()
            for w in self.module.parameters():          # every [N, K] matrix: nn.Linear weights, in_proj_weight, 1x1x1 convolutions
                if not (w.dim() == 2 or (w.dim() == 5 and tuple(w.shape[1:4]) == (1, 1, 1))):
                    continue
                if not w.is_cuda or not w.requires_grad or w.data_ptr() in seen or not w.is_contiguous():
                    continue
                N, K = w.shape[0], w.numel() // w.shape[0]
                if N % 16 == 0 and K % 4 == 0:
                    seen.add(w.data_ptr())
                    ws.append((w, N, K))
        if ws:
            dev = ws[0][0].device
            flat = torch.empty(sum(N * K for _, N, K in ws), dtype=torch.float32, device=dev)
            rows, blocks, off, table, vers = [], 0, 0, {}, {}
            for w, N, K in ws:
                wt = flat[off:off + N * K].view(K, N)
                off += N * K
                rows.append([w.data_ptr(), wt.data_ptr(), N, K, blocks])
                blocks += ((N + 31) // 32) * ((K + 31) // 32)
                table[w.data_ptr()] = wt
                vers[w.data_ptr()] = w._version
            desc = L.h2d(rows, torch.int64, dev)
            L.call('u3d_transpose_batch', L.ptr(desc), len(rows), blocks, L.stream())
            table['_keep'] = (flat, desc)
            table['_versions'] = vers
            if _W_PLANES and not P.bf16() and P.get_fp32_math() == 'bf16x3':
                # three-plane products: the planes of every weight AND of its transposed copy, one launch (u3d_weight_planes_batch);
                # planes[(data_ptr of the fp32 matrix)] -> bf16 [3, rows, cols]
                mats = [(w, N * K) for w, N, K in ws] + [(table[w.data_ptr()], N * K) for w, N, K in ws]
                pflat = torch.empty(3 * sum(n for _, n in mats), dtype=torch.bfloat16, device=dev)
                prow, pblocks, poff, planes = [], 0, 0, {}
                for m, n in mats:
                    pl = pflat[poff:poff + 3 * n]
                    poff += 3 * n
                    prow.append([m.data_ptr(), pl.data_ptr(), n // 8, pblocks])
                    pblocks += (n // 8 + 255) // 256
                    planes[m.data_ptr()] = pl
                pdesc = L.h2d(prow, torch.int64, dev)
                L.call('u3d_weight_planes_batch', L.ptr(pdesc), len(prow), pblocks, L.stream())
                table['_planes'] = planes
                table['_keep'] += (pflat, pdesc)
            _WT_ACTIVE = table
        else:
            _WT_ACTIVE = None
        return self

    def __exit__(self, *exc):
        global _WT_ACTIVE
        _WT_ACTIVE = self.prev
        return False


# Pre-split W operands of the three-plane NT products (u3d_weight_planes_batch + u3d_gemm_w_planes): built, bit-identical, and OFF by
# default -- measured on MI355X (round 5, same box, two runs each): all GEMMs 5.40 / 5.34 ms with the planes, 5.41 / 5.41 without, step
# 293.5 / 292.5 vs 294.8 / 295.7 scenes/s.  The W split is a third of the NT kernel's split arithmetic (loop VALU 156 -> 110
# instructions, tools/isa_mix.py) and removing it buys nothing: the kernel is not bound by its VALU issue.  U3D_W_PLANES=1 turns it on.
_W_PLANES = os.environ.get('U3D_W_PLANES', '0') == '1'


def _planes_of(mat):
    """bf16 [3 * N * K] planes of an fp32 matrix (a weight or its transposed copy) made by the enclosing transposed_weights()
    context; None outside one.  Callers keep the tensor (ctx) for as long as a launch may read it."""
    if _WT_ACTIVE is None or mat is None:
        return None
    return _WT_ACTIVE.get('_planes', {}).get(mat.data_ptr())


def _use_planes(w1, p1, w2=None, p2=None):
    """hand the next NT launch its pre-split W operand(s) together with the matrices they belong to (include/u3d.h
    u3d_gemm_w_planes: the launch ignores planes of any other matrix); no call when there are none"""
    if p1 is not None or p2 is not None:
        L.lib().u3d_gemm_w_planes(L.ptr(w1) if p1 is not None else None, L.ptr(p1), L.ptr(w2) if p2 is not None else None, L.ptr(p2))


def _wt_of(weight):
    """the [K, N] copy of ``weight`` made by the enclosing transposed_weights() context (None outside one / for other tensors)"""
    if _WT_ACTIVE is None:
        return None
    wt = _WT_ACTIVE.get(weight.data_ptr())
    if wt is None or wt.shape != (weight.numel() // weight.shape[0], weight.shape[0]):
        return None
    # a parameter modified in place since the context was entered (EMA update, in-forward optimizer step): its copy is
    # stale -> None, and the op transposes for itself
    return wt if _WT_ACTIVE.get('_versions', {}).get(weight.data_ptr()) == weight._version else None


def _flops(M, N, K, extra_mn=0):
    """algorithmic flops of an [M,K] x [K,N] product, booked with its bytes (operands + result (+ extra [M,N] streams))"""
    if not _PROFILE_FLOPS:
        return 0.0
    account.add('gemm', 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N * (1 + extra_mn)))
    return 2.0 * M * N * K


def _input_grad(dy, weight, act=ACT_NONE, aux=None, bf=False, wt=None, wt_planes=None):
    """dX[M,K] = dY[M,N] . W[N,K], optionally times act'(aux) in the GEMM epilogue (u3d_linear_dact): the input gradient
    THROUGH the activation that produced this layer's input (aux = its ReLU output / GELU pre-activation)."""
    M, N = dy.shape
    K = weight.shape[1]
    dev = dy.device
    q = 32 if bf else 16                                # reduction-depth granule of the kernel
    if N % q == 0:
        if wt is None:              # (``wt``: the copy transposed_weights() made for this forward pass; ``wt_planes``: its planes)
            wt = torch.empty(K, N, dtype=torch.float32, device=dev)
            L.call('u3d_transpose', L.ptr(weight.contiguous()), L.ptr(wt), N, K, L.stream())
            wt_planes = None
    else:
        wt_planes = None                                               # tiny heads (N = 19, 8): pad the reduction dim with zero columns
        Np = (N + q - 1) // q * q
        wt = torch.zeros(K, Np, dtype=torch.float32, device=dev)
        wt[:, :N] = weight.t()
        dyp = torch.zeros(M, Np, dtype=torch.float32, device=dev)
        dyp[:, :N] = dy
        dy, N = dyp, Np
    if act == ACT_NONE:
        return _gemm_nt(dy, wt, None, bf, None if bf else wt_planes)
    dx = torch.empty(M, K, dtype=torch.float32, device=dev)
    if M:
        if not bf:
            _use_planes(wt, wt_planes)
        L.call('u3d_linear_dact', L.ptr(dy), L.ptr(wt), L.ptr(aux), act | (P.BF16_FLAG if bf else 0), L.ptr(dx), M, K, N,
               _flops(M, K, N, extra_mn=1), L.stream())
    return dx


def _weight_grad(dy, x, want_bias, bf=False):
    """(dW[N,K] = dY^T X, db[N] = column sums of dY or None): one pass of u3d_gemm_tn (+ fixed-order reduce)."""
    tn = 'u3d_gemm_tn_bf16' if bf else 'u3d_gemm_tn'
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    db = None
    dw = torch.empty(N, K, dtype=torch.float32, device=dev)
    if M and N % 4 == 0:
        ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, N, K), dev)
        if want_bias:                                   # the bias gradient (column sums of dy) rides along
            db = torch.empty(N, dtype=torch.float32, device=dev)
        L.call(tn, L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), M, N, K, L.ptr(ws), _flops(M, N, K), L.stream())
    elif M:
        Np = (N + 3) // 4 * 4
        dyp = torch.zeros(M, Np, dtype=torch.float32, device=dev)
        dyp[:, :N] = dy
        dwp = torch.empty(Np, K, dtype=torch.float32, device=dev)
        ws = L.scratch(L.lib().u3d_gemm_tn_ws_bytes(M, Np, K), dev)
        dbp = torch.empty(Np, dtype=torch.float32, device=dev) if want_bias else None
        L.call(tn, L.ptr(dyp), L.ptr(x), L.ptr(dwp), L.ptr(dbp), M, Np, K, L.ptr(ws), 0.0, L.stream())
        dw = dwp[:N].contiguous()
        if dbp is not None:
            db = dbp[:N].contiguous()
    else:
        dw.zero_()
        if want_bias:
            db = torch.zeros(N, dtype=torch.float32, device=dev)
    return dw, db


def _weight_grad_overlapped(dy, x, want_bias, bf, weight, bias):
    """``_weight_grad`` on the weight-gradient side stream when sparse.set_wgrad_overlap(2) is on (nothing downstream of a Linear needs
    its dW: the GEMM joins the sparse convolutions' weight-gradient chain and the dX chain goes on without it).  Only for leaf
    parameters without an existing .grad -- autograd then just stores the tensor; anything else is computed in line."""
    from . import sparse
    ok = sparse._WGRAD_OVERLAP == 2 and _OVERLAP_TN and dy.is_cuda and sparse.async_dw_ok(weight, bias)
    if not ok:
        return _weight_grad(dy, x, want_bias, bf)
    dev = dy.device
    main = torch.cuda.current_stream(dev)
    side = sparse._side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw, db = _weight_grad(dy, x, want_bias, bf)
    dy.record_stream(side)
    x.record_stream(side)
    for t in (dw, db):                      # allocated from the side stream's pool, consumed on the main stream after the join
        if t is not None:
            t.record_stream(main)
    sparse._queue_join(dev)
    return dw, db


_OVERLAP_TN = os.environ.get('U3D_OVERLAP_TN', '1') != '0'


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.bf = P.bf16()
        ctx.wt = _wt_of(weight)
        ctx.wt_planes = _planes_of(ctx.wt)
        return _gemm_nt(x, weight.contiguous(), bias, ctx.bf, None if ctx.bf else _planes_of(weight))

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[1]:           # first: on the side stream it then waits for dy only, not for the dX product below
            dw, db = _weight_grad_overlapped(dy, x, ctx.has_bias and ctx.needs_input_grad[2], ctx.bf, weight, ctx.bias_ref)
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy, weight, bf=ctx.bf, wt=ctx.wt, wt_planes=ctx.wt_planes)
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def _act16(x, *reduction_dims):
    """the bf16-activation data flow of dense16.py applies: precision.bf16_act(), a device tensor, every reduction depth % 32 == 0"""
    return P.bf16_act() and x.is_cuda and all(d % 32 == 0 for d in reduction_dims)


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None, out_bf16: bool = False) -> torch.Tensor:
    """y = x W^T + b for 2-D x [M, K] (K % 16 == 0).  ``out_bf16``: under precision.bf16_act() the result is a bf16 tensor (for a
    consumer that reads one: the attention kernels); ignored otherwise."""
    if _act16(x, x.shape[1]):
        from . import dense16
        return dense16.linear(x, weight, bias, out_bf16)
    return _LinearFn.apply(x, weight, bias)


# GELU in the GEMM epilogues (True) or as stand-alone passes u3d_gelu_fwd / u3d_gelu_bwd between plain GEMMs (False).  ReLU is
# always fused (one v_max / one compare per element).  Measured on MI355X (tools/prof_mlp.py, FFN 256 -> 1024 -> 256 over 16.8 k
# rows, forward + backward): fused 0.761 ms, stand-alone passes 0.682 ms (bf16 operands: 0.487 / 0.438) -- the ~15 VALU
# instructions per element of the erf GELU are not hidden in a GEMM epilogue (one wave per SIMD), while the stand-alone pass
# runs at HBM speed; so the default is the stand-alone pass.
FUSE_GELU = os.environ.get('U3D_FUSE_GELU', '0') == '1'


class _MLPFn(torch.autograd.Function):
    """z = act(x W1^T + b1) W2^T + b2 (include/u3d.h u3d_ffn_fwd): bias and activation live in the first GEMM's epilogue, the
    activation's derivative in the epilogue of the GEMM that produces the hidden gradient -- no elementwise kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act):
        x = x.contiguous()
        M, d_in = x.shape
        hid, d_out = w1.shape[0], w2.shape[0]
        dev = x.device
        a = torch.empty(M, hid, dtype=torch.float32, device=dev)
        h = torch.empty(M, hid, dtype=torch.float32, device=dev) if act == ACT_GELU else None
        z = torch.empty(M, d_out, dtype=torch.float32, device=dev)
        w1c, w2c = w1.contiguous(), w2.contiguous()
        ctx.bf = P.bf16()
        ctx.wt1, ctx.wt2 = _wt_of(w1c), _wt_of(w2c)
        ctx.wtp1, ctx.wtp2 = _planes_of(ctx.wt1), _planes_of(ctx.wt2)
        p1, p2 = (None, None) if ctx.bf else (_planes_of(w1c), _planes_of(w2c))
        ctx.fused = act != ACT_GELU or FUSE_GELU
        if M and not ctx.fused:                      # plain GEMM (+bias) -> GELU pass -> plain GEMM
            h = _gemm_nt(x, w1c, b1, ctx.bf, p1)
            L.call('u3d_gelu_fwd', L.ptr(h), L.ptr(a), M * hid, L.stream())
            z = _gemm_nt(a, w2c, b2, ctx.bf, p2)
        elif M:
            _flops(M, hid, d_in, extra_mn=1 if act == ACT_GELU else 0)
            _flops(M, d_out, hid)
            _use_planes(w1c, p1, w2c, p2)
            L.call('u3d_ffn_fwd', L.ptr(x), L.ptr(w1c), L.ptr(b1), L.ptr(w2c), L.ptr(b2), act | (P.BF16_FLAG if ctx.bf else 0),
                   L.ptr(h), L.ptr(a), L.ptr(z), M, d_in, hid, d_out, 1.0 if _PROFILE_FLOPS else 0.0, L.stream())
        ctx.save_for_backward(x, w1c, w2c, a, h)
        ctx.act, ctx.bias = act, (b1 is not None, b2 is not None)
        ctx.bias_refs = (b1, b2)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w1, w2, a, h = ctx.saved_tensors
        dz = dz.contiguous()
        need = ctx.needs_input_grad
        dw2, db2 = _weight_grad_overlapped(dz, a, ctx.bias[1] and need[4], ctx.bf, w2, ctx.bias_refs[1]) if need[3] else (None, None)
        if ctx.fused:
            dh = _input_grad(dz, w2, ctx.act, h if ctx.act == ACT_GELU else a, bf=ctx.bf, wt=ctx.wt2, wt_planes=ctx.wtp2)
        else:
            da = _input_grad(dz, w2, bf=ctx.bf, wt=ctx.wt2, wt_planes=ctx.wtp2)
            dh = torch.empty_like(da)
            if da.numel():
                L.call('u3d_gelu_bwd', L.ptr(da), L.ptr(h), L.ptr(dh), da.numel(), L.stream())
        dw1, db1 = _weight_grad_overlapped(dh, x, ctx.bias[0] and need[2], ctx.bf, w1, ctx.bias_refs[0]) if need[1] else (None, None)
        dx = _input_grad(dh, w1, bf=ctx.bf, wt=ctx.wt1, wt_planes=ctx.wtp1) if need[0] else None
        return dx, dw1, db1, dw2, db2, None


def mlp(x, w1, b1, w2, b2, act: str) -> torch.Tensor:
    """Linear -> ReLU / GELU -> Linear on 2-D x (d_in, hidden % 16 == 0)."""
    if _act16(x, x.shape[1], w1.shape[0]):
        from . import dense16
        return dense16.mlp(x, w1, b1, w2, b2, {'relu': ACT_RELU, 'gelu': ACT_GELU}[act])
    return _MLPFn.apply(x, w1, b1, w2, b2, {'relu': ACT_RELU, 'gelu': ACT_GELU}[act])


def _aliases(y, n_out):
    """``y`` as ``n_out`` autograd outputs that share its storage: one per consumer, so that their gradients arrive separately at the
    producing Function's backward -- which sums them while it reads them -- instead of being added by autograd's elementwise kernels"""
    return y if n_out == 1 else (y,) + tuple(y.view_as(y) for _ in range(n_out - 1))


def _grads_in(dys):
    """the incoming gradients of ``_aliases`` outputs: (first, second or None, third or None) of those that arrived"""
    g = [d.contiguous() for d in dys if d is not None]
    if len(g) > 3:
        raise L.U3DError('layer_norm: at most three consumers of the result are summed in the backward kernel')
    return g + [None] * (3 - len(g))


class _LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res) over the last dimension (include/u3d.h K15); the gradient of x and res is the same tensor.  ``n_out`` > 1:
    the result is returned that many times (one output per consumer) and the backward kernel sums their gradients (u3d_layer_norm_bwd_sum)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps, n_out=1):
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
        ctx.set_materialize_grads(False)
        return _aliases(y, n_out)

    @staticmethod
    def backward(ctx, *dys):
        s, weight, stats = ctx.saved_tensors
        dy, dy2, dy3 = _grads_in(dys)
        if dy is None:
            return (None,) * 6
        M, C = s.shape
        dx = torch.empty_like(s)
        dg = torch.empty(C, dtype=torch.float32, device=s.device)
        db = torch.empty(C, dtype=torch.float32, device=s.device)
        if M:
            ws = L.scratch(L.lib().u3d_layer_norm_ws_bytes(M, C), s.device)
            L.call('u3d_layer_norm_bwd_sum', L.ptr(s), L.ptr(dy), L.ptr(dy2), L.ptr(dy3), L.ptr(weight), L.ptr(stats), M, C, L.ptr(dx), None,
                   L.ptr(dg), L.ptr(db), L.ptr(ws), L.stream())
        else:
            dg.zero_(); db.zero_()
        return dx, (dx if ctx.has_res else None), dg, db, None, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, res: torch.Tensor = None, n_out: int = 1):
    """LayerNorm(x + res) for 2-D x [M, C] (C % 4 == 0, C <= 1024).  ``n_out`` in (2, 3): a tuple of that many tensors, all THE result
    (shared storage) -- hand each consumer its own and the backward kernel sums their gradients (no elementwise add passes)."""
    if _act16(x, x.shape[1]):
        from . import dense16
        return dense16.layer_norm(x, weight, bias, eps, res, n_out)
    return _LayerNormFn.apply(x, res, weight, bias, eps, n_out)


class _LNLinearFn(torch.autograd.Function):
    """(nq, y) = (LayerNorm(x), nq W^T + b)  -- include/u3d.h u3d_ln_linear.  Backward folds the two gradient contributions of
    ``nq`` (its own consumers + this Linear) into the GEMM that produces the second (u3d_gemm_nt_add), then runs the
    LayerNorm backward once."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, weight, bias):
        x = x.contiguous()
        M, C = x.shape
        N = weight.shape[0]
        dev = x.device
        nq = torch.empty_like(x)
        stats = torch.empty(M, 2, dtype=torch.float32, device=dev)
        y = torch.empty(M, N, dtype=torch.float32, device=dev)
        w = weight.contiguous()
        ctx.bf = P.bf16() and C % 32 == 0
        if M:
            L.call('u3d_ln_linear', L.ptr(x), None, L.ptr(gamma), L.ptr(beta), float(eps), None, L.ptr(nq), L.ptr(stats), L.ptr(w),
                   L.ptr(bias), P.BF16_FLAG if ctx.bf else 0, None, L.ptr(y), M, C, N, _flops(M, N, C), L.stream())
        ctx.save_for_backward(x, gamma, stats, nq, w)
        ctx.has_bias = bias is not None
        return nq, y

    @staticmethod
    def backward(ctx, dnq, dy):
        x, gamma, stats, nq, w = ctx.saved_tensors
        M, C = x.shape
        N = w.shape[0]
        dev = x.device
        dw = db = None
        if dy is not None:
            dy = dy.contiguous()
            dw, db = _weight_grad(dy, nq, ctx.has_bias, ctx.bf)
            if dnq is None:
                dtot = _input_grad(dy, w, bf=ctx.bf)
            else:                                        # dtot = dy W + dnq in one GEMM
                q = 32 if ctx.bf else 16
                Np = (N + q - 1) // q * q
                wt = torch.zeros(C, Np, dtype=torch.float32, device=dev)
                wt[:, :N] = w.t()
                dyp = dy if Np == N else torch.nn.functional.pad(dy, (0, Np - N))
                dtot = torch.empty(M, C, dtype=torch.float32, device=dev)
                if M:
                    L.call('u3d_gemm_nt_add', L.ptr(dyp), L.ptr(wt), L.ptr(dnq.contiguous()), P.BF16_FLAG if ctx.bf else 0, L.ptr(dtot),
                           M, C, Np, _flops(M, C, Np, extra_mn=1), L.stream())
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


def ln_linear(x, gamma, beta, eps, weight, bias):
    """-> (LayerNorm(x), LayerNorm(x) W^T + b) for 2-D x [M, C]."""
    if _act16(x, x.shape[1]):
        from . import dense16
        return dense16.ln_linear(x, gamma, beta, eps, weight, bias)
    return _LNLinearFn.apply(x, gamma, beta, eps, weight, bias)


class LayerNorm(torch.nn.LayerNorm):
    """``nn.LayerNorm(d_model)`` of the reference (same parameters / state_dict keys) on the HIP kernels, optionally fused with the
    residual add in front of it."""

    def forward(self, x, res=None, n_out=1):
        return layer_norm(x, self.weight, self.bias, self.eps, res, n_out)

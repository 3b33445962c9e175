"""ctypes binding of the C-ABI kernel library (include/u3d.h).

The product path has NO fallback: if ``libu3d_hip.so`` is missing or a kernel
returns an error, an exception is raised -- nothing here routes to PyTorch
ops or to the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('U3D_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libu3d_hip.so')      # override: A/B runs of kernel builds

_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes)   (mirrors include/u3d.h, checked by tests/test_cabi.py)
PROTOTYPES = {
    'u3d_version': (_i32, []),
    'u3d_last_error': (C.c_char_p, []),
    'u3d_fp32_math': (_i32, [_i32]),
    'u3d_conv_kernel': (_i32, [_i32]),
    'u3d_prof_enable': (_i32, [_i32, _i32]),
    'u3d_prof_collect': (_i32, [_i32, C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_f64)]),
    'u3d_vox_scene_stats': (_i32, [_vp, _vp, _vp, _i32, _i64, _f32, _i32, _vp, _vp, _vp, _vp]),
    'u3d_vox_scene_stats_ws_bytes': (_i64, [_i32]),
    'u3d_index_words': (_i64, [_i32, _i32, _i32, _i32]),
    'u3d_vox_mark': (_i32, [_vp, _vp, _vp, _i32, _i64, _vp, _f32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    'u3d_index_rank': (_i32, [_vp, _i64, _vp, _vp, _vp]),
    'u3d_index_rank_ws_bytes': (_i64, [_i64]),
    'u3d_index_coords': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'u3d_vox_finalize': (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    'u3d_hash_index_slots': (_i64, [_i64]),
    'u3d_hash_index_ws_bytes': (_i64, [_i64]),
    'u3d_hash_index_build': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'u3d_sort_u64': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    'u3d_sort_ws_bytes': (_i64, [_i64, _i32]),
    'u3d_hash_index_coords': (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    'u3d_cells_of_coords': (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'u3d_vox_finalize_ws_bytes': (_i64, [_i64, _i64]),
    'u3d_subm_rulebook': (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'u3d_subm_rulebook_ws_bytes': (_i64, [_i64]),
    'u3d_index_mark': (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    'u3d_down_rulebook': (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'u3d_down_rulebook_ws_bytes': (_i64, [_i64]),
    'u3d_tile_starts': (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp]),
    'u3d_spconv_gmm': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _f64, _vp]),
    'u3d_spconv_gmm_bf16': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _f64, _vp]),
    'u3d_spconv_gmm_bf16a': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _f64, _vp]),
    'u3d_weight_pack_batch': (_i32, [_vp, _i32, _i64, _vp]),
    'u3d_weight_pack_bf16': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'u3d_spconv_gmm_x3': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _f64, _vp]),
    'u3d_weight_pack_x3': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'u3d_spconv_rs_x3': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f64, _vp]),
    'u3d_spconv_rs_bf16a': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f64, _vp]),
    'u3d_spconv_ts_plan': (_i32, [_i32, _i32, _i64, C.POINTER(_i32), C.POINTER(_i32)]),
    'u3d_subm_halo_pmax': (_i32, [_i32, _i32]),
    'u3d_subm_halo': (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'u3d_spconv_ts_x3': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f64, _vp]),
    'u3d_spconv_plan': (_i32, [_i32, _i32, _i32, _i64, C.POINTER(_i32), C.POINTER(_i32)]),
    'u3d_spconv_plan_bf16a': (_i32, [_i32, _i32, _i32, _i64, C.POINTER(_i32), C.POINTER(_i32)]),
    'u3d_spconv_wgrad': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _f64, _vp]),
    'u3d_spconv_wgrad_bf16': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _f64, _vp]),
    'u3d_spconv_wgrad_rows': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _f64, _vp]),
    'u3d_spconv_wgrad_rows_supported': (_i32, [_i32, _i32]),
    'u3d_spconv_wgrad_tile_rows': (_i32, [_i32, _i64, _i32, _i32]),
    'u3d_spconv_wgrad_ws_bytes': (_i64, [_i32, _i64, _i32, _i32]),
    'u3d_layer_norm_fwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    'u3d_layer_norm_bwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    'u3d_layer_norm_ws_bytes': (_i64, [_i64, _i32]),
    'u3d_layer_norm_fwd_b16': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    'u3d_layer_norm_bwd_b16': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'u3d_layer_norm_bwd_sum': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'u3d_nms_bev': (_i32, [_vp, _vp, _i32, _f32, _vp, _vp]),
    'u3d_nms_aligned3d': (_i32, [_vp, _vp, _i32, _f32, _vp, _vp]),
    'u3d_nms_rotated': (_i32, [_vp, _vp, _i32, _f32, _vp, _vp]),
    'u3d_trim_boxes': (_i32, [_vp, _i64, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _f32, _vp, _vp]),
    'u3d_weight_pack': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'u3d_weight_transpose': (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    'u3d_bn_stats': (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp]),
    'u3d_bn_ws_bytes': (_i64, [_i32]),
    'u3d_bn_forward': (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'u3d_bn_backward': (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'u3d_bn_finalize': (_i32, [_vp, _f64, _vp, _vp, _f32, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'u3d_bn_apply': (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    'u3d_bn_bwd_stats': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    'u3d_bn_bwd_apply': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _f64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'u3d_csr_build': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    'u3d_csr_build_ws_bytes': (_i64, [_i64, _i64]),
    'u3d_gather_i64_to_i32': (_i32, [_vp, _vp, _i64, _vp, _vp]),
    'u3d_segment_gather_sum': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    'u3d_segment_mean_xyz': (_i32, [_vp, _i32, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp]),
    'u3d_segment_minmax_xyz': (_i32, [_vp, _i32, _vp, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    'u3d_criterion_packed': (_i32, [_vp] * 11 + [_i32, _i32, _i64, _i32, _i32, _i64, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    'u3d_box_decode_fwd': (_i32, [_vp, _vp, _i64, _vp, _vp]),
    'u3d_box_decode_bwd': (_i32, [_vp, _vp, _i64, _vp, _vp]),
    'u3d_box_decode7_fwd': (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    'u3d_box_decode7_bwd': (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    'u3d_criterion_ws_bytes': (_i64, [_i32, _i32, _i64, _i64, _i64]),
    'u3d_gemm_nt': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f64, _vp]),
    'u3d_linear_act': (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _f64, _vp]),
    'u3d_linear_dact': (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f64, _vp]),
    'u3d_gemm_nt_add': (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f64, _vp]),
    'u3d_ln_linear': (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _f64, _vp]),
    'u3d_ffn_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f64, _vp]),
    'u3d_gelu_fwd': (_i32, [_vp, _vp, _i64, _vp]),
    'u3d_gelu_bwd': (_i32, [_vp, _vp, _vp, _i64, _vp]),
    'u3d_gemm_nt_b16': (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i64, _i32, _i32, _f64, _vp]),
    'u3d_gemm_tn_b16': (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _f64, _vp]),
    'u3d_gemm_tn_b16_ws_bytes': (_i64, [_i64, _i32, _i32]),
    'u3d_gelu_fwd_b16': (_i32, [_vp, _vp, _i64, _vp]),
    'u3d_gelu_bwd_b16': (_i32, [_vp, _vp, _vp, _i64, _vp]),
    'u3d_gemm_tn': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _f64, _vp]),
    'u3d_gemm_tn_bf16': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _f64, _vp]),
    'u3d_gemm_tn_ws_bytes': (_i64, [_i64, _i32, _i32]),
    'u3d_transpose': (_i32, [_vp, _vp, _i32, _i32, _vp]),
    'u3d_transpose_batch': (_i32, [_vp, _i32, _i64, _vp]),
    'u3d_weight_planes_batch': (_i32, [_vp, _i32, _i64, _vp]),
    'u3d_gemm_w_planes': (_i32, [_vp, _vp, _vp, _vp]),
    'u3d_attn_varlen_fwd': (_i32, [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _f64, _vp]),
    'u3d_attn_varlen_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _f64, _vp]),
    'u3d_attn_varlen_fwd_bf16': (_i32, [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _f64, _vp]),
    'u3d_attn_varlen_bwd_bf16': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _f64, _vp]),
    'u3d_attn_varlen_fwd_b16': (_i32, [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _f64, _vp]),
    'u3d_attn_varlen_bwd_b16': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _f64, _vp]),
}

ABI_VERSION = 114         # include/u3d.h U3D_ABI_VERSION this table was written against

K_CONV_FWD, K_CONV_WGRAD, K_BN, K_POOL, K_ATTN_FWD, K_ATTN_BWD, K_RULEBOOK, K_VOXELIZE, K_GEMM = range(9)

_lib = None


class U3DError(RuntimeError):
    pass


def lib():
    """Load the kernel library (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise U3DError(f'{LIB_PATH} not found: build it with `python -m unidet3d_amd.csrc.build` '
                           '(hipcc --offload-arch=gfx950); there is no fallback path')
        l = C.CDLL(LIB_PATH)
        # version first: a stale library that lacks a newer symbol must say "rebuild", not raise AttributeError from the loop below
        l.u3d_version.restype = C.c_int
        v = l.u3d_version() if hasattr(l, 'u3d_version') else -1
        if v != ABI_VERSION:
            raise U3DError(f'{LIB_PATH} has ABI version {v}, this package expects {ABI_VERSION}: rebuild it '
                           '(`python -m unidet3d_amd.csrc.build`); a stale library would misread the arguments')
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def call(name: str, *args):
    """Call a status-returning entry point; raise on any non-zero code."""
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise U3DError(f'{name} failed with code {rc}: {l.u3d_last_error().decode()}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Tensors must be contiguous CUDA tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise U3DError('u3d kernels need CUDA (HIP) tensors: the product path has no CPU fallback')
    if not t.is_contiguous():
        raise U3DError('non-contiguous tensor passed to a u3d kernel')
    return t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream (the private accessor costs ~0.3 us; torch.cuda.current_stream() builds a
    Stream object and resolves the device through three Python layers, ~10 us -- a thousand launches per step pay it)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def h2d(values, dtype, device) -> torch.Tensor:
    """Small host list -> device tensor without draining the stream: pinned staging + non_blocking copy
    (a pageable H2D copy makes the host wait for everything already queued on the stream)."""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def h2d_pack(specs, device):
    """Several small host lists -> device tensors through ONE pinned staging buffer and ONE non-blocking copy.  ``specs``: [(values,
    dtype), ...] (nested lists are flattened row-major); returns the 1-D device tensors in order (views of one buffer, 16-byte aligned)."""
    parts, off = [], 0
    for values, dtype in specs:
        t = torch.tensor(values, dtype=dtype).reshape(-1)
        parts.append((t, off))
        off = (off + t.numel() * t.element_size() + 15) // 16 * 16
    host = torch.empty(max(off, 16), dtype=torch.uint8)
    if torch.device(device).type == 'cuda':
        host = host.pin_memory()
    for t, o in parts:
        if t.numel():
            host[o:o + t.numel() * t.element_size()].view(t.dtype).copy_(t)
    dev = host.to(device, non_blocking=True)
    return [dev[o:o + t.numel() * t.element_size()].view(t.dtype) for t, o in parts]


_SCRATCH = {}


def scratch(nbytes: int, device) -> torch.Tensor:
    """Persistent workspace per (device, stream): reuse is ordered by the stream it is used on, so work queued on a side stream
    (UniDet3D.prefetch) never shares a buffer with the kernels of the main stream."""
    key = (device, stream())
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 21), dtype=torch.uint8, device=device)
        _SCRATCH[key] = t
    return t


def prof_enable(cls: int, on: bool):
    call('u3d_prof_enable', cls, 1 if on else 0)


def prof_collect(cls: int):
    ms, n, w = _f64(), _i64(), _f64()
    call('u3d_prof_collect', cls, C.byref(ms), C.byref(n), C.byref(w))
    return ms.value, n.value, w.value

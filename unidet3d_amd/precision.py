"""MFMA operand precision of the hot path (BASELINE configs[1] vs configs[2]).

``fp32`` (default): fp32 operands and fp32-accurate products -- the reference's arithmetic without ``--amp``; the headline bench
line.  HOW the fp32 products are formed is a second, library-wide switch (``fp32_math``): ``'bf16x3'`` (default) splits every
operand exactly into three bf16 pieces and runs six bf16 MFMAs per product (error of an fp32 FMA chain, 2.7x less matrix-pipe
time), ``'mfma'`` uses the native v_mfma_f32_*_f32 instructions (include/u3d.h: u3d_fp32_math; env U3D_FP32_MATH).
``bf16``: feature / weight / probability operands are rounded to bf16 (round to nearest even) inside the kernels, on their way
into LDS or registers, and multiplied on v_mfma_f32_*_bf16; tensors stay fp32 in HBM, accumulators, batch-norm statistics,
softmax statistics, LayerNorm, the loss and the optimizer stay fp32 -- the mixed-precision recipe of the reference's
``--amp`` switch (tools/train.py:86-99: AmpOptimWrapper around the same model) with bf16 instead of fp16 operands.

The mode is read when an op runs forward and is remembered for that op's backward.
"""
from __future__ import annotations

import contextlib
import os

_MODE = 'fp32'
BF16_FLAG = 16          # include/u3d.h U3D_BF16_OPERANDS


def set_operand_dtype(mode: str):
    global _MODE
    if mode not in ('fp32', 'bf16'):
        raise ValueError("operand dtype must be 'fp32' or 'bf16'")
    _MODE = mode


def operand_dtype() -> str:
    return _MODE


def bf16() -> bool:
    return _MODE == 'bf16'


@contextlib.contextmanager
def operands(mode: str):
    prev = _MODE
    set_operand_dtype(mode)
    try:
        yield
    finally:
        set_operand_dtype(prev)


_FP32_MATH = {'mfma': 0, 'bf16x3': 1}
_X3 = None              # cached library state (u3d_fp32_math is process-wide)


def get_fp32_math() -> str:
    global _X3
    if _X3 is None:
        from . import _lib as L
        _X3 = L.lib().u3d_fp32_math(-1) == 1
    return 'bf16x3' if _X3 else 'mfma'


def set_fp32_math(mode: str) -> str:
    """'bf16x3' | 'mfma' (see the module docstring); returns the previous mode.  Process-wide."""
    global _X3
    from . import _lib as L
    if mode not in _FP32_MATH:
        raise ValueError("fp32 math must be 'bf16x3' or 'mfma'")
    prev = 'bf16x3' if L.lib().u3d_fp32_math(_FP32_MATH[mode]) == 1 else 'mfma'
    _X3 = mode == 'bf16x3'
    return prev


@contextlib.contextmanager
def fp32_math(mode: str):
    prev = set_fp32_math(mode)
    try:
        yield
    finally:
        set_fp32_math(prev)


_CONV_KERNEL = {'wave': 0, 'workgroup': 1, 'workgroup-all': 2}


def set_conv_kernel(kind: str) -> str:
    """'workgroup' (default: four waves share a row tile and the LDS copy of an offset's weights, csrc/spconv_wg.hip, where that
    measured ahead: three-plane products, single offset group) | 'workgroup-all' (wherever the kernel is instantiated) | 'wave'
    (wave-private tiles, csrc/spconv.hip) for the bf16 / three-plane sparse convolutions (include/u3d.h: u3d_conv_kernel;
    env U3D_GMM_WG).  Returns the previous kind.  Process-wide; A/B measurements and tests."""
    from . import _lib as L
    if kind not in _CONV_KERNEL:
        raise ValueError("conv kernel must be 'workgroup', 'workgroup-all' or 'wave'")
    return {0: 'wave', 1: 'workgroup', 2: 'workgroup-all'}[L.lib().u3d_conv_kernel(_CONV_KERNEL[kind])]


@contextlib.contextmanager
def conv_kernel(kind: str):
    prev = set_conv_kernel(kind)
    try:
        yield
    finally:
        set_conv_kernel(prev)


FMT_FP32, FMT_BF16, FMT_X3 = 0, 1, 2

# bf16 operands (BASELINE configs[2]): batch-norm outputs and the gradients a batch norm hands back are ALSO written as bf16 rows
# (a shadow next to the fp32 tensor, sparse.attach_shadow) and the sparse convolutions gather those -- half the gathered bytes, the
# MFMA fragment straight from the row.  Bit-identical to gathering the fp32 rows and rounding them (what FMT_BF16 does); off:
# U3D_BF16_ROWS=0 / set_bf16_rows(False).
_BF16_ROWS = os.environ.get('U3D_BF16_ROWS', '1') != '0'


def set_bf16_rows(on: bool) -> bool:
    global _BF16_ROWS
    prev, _BF16_ROWS = _BF16_ROWS, bool(on)
    return prev


def bf16_rows() -> bool:
    """True when bf16-operand mode also keeps bf16 copies of the rows the sparse convolutions gather."""
    return _MODE == 'bf16' and _BF16_ROWS


@contextlib.contextmanager
def bf16_rows_mode(on: bool):
    prev = set_bf16_rows(on)
    try:
        yield
    finally:
        set_bf16_rows(prev)


# bf16 operands (BASELINE configs[2]), decoder side: the activations between the decoder's Linear layers ARE bf16 tensors in HBM
# (dense16.py, include/u3d.h K14b) -- what the reference's autocast does to nn.Linear / nn.MultiheadAttention -- instead of fp32
# tensors rounded in flight.  Off: U3D_BF16_ACT=0 / set_bf16_act(False) (the round-5 data flow).
_BF16_ACT = os.environ.get('U3D_BF16_ACT', '1') != '0'


def set_bf16_act(on: bool) -> bool:
    global _BF16_ACT
    prev, _BF16_ACT = _BF16_ACT, bool(on)
    return prev


def bf16_act() -> bool:
    """True when bf16-operand mode also keeps the decoder's activations in bf16."""
    return _MODE == 'bf16' and _BF16_ACT


@contextlib.contextmanager
def bf16_act_mode(on: bool):
    prev = set_bf16_act(on)
    try:
        yield
    finally:
        set_bf16_act(prev)


def conv_format() -> int:
    """Operand format of the sparse-convolution kernels for the current modes: the packed-weight layout and the entry point
    (u3d_spconv_gmm / _bf16 / _x3) go together, so the choice is made here, once per op, and remembered for its backward."""
    if _MODE == 'bf16':
        return FMT_BF16
    return FMT_X3 if get_fp32_math() == 'bf16x3' else FMT_FP32

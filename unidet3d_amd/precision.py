"""MFMA operand precision of the hot path (BASELINE configs[1] vs configs[2]).

``fp32`` (default): v_mfma_f32_*_f32 everywhere -- the reference's arithmetic without ``--amp``; the headline bench line.
``bf16``: feature / weight / probability operands are rounded to bf16 (round to nearest even) inside the kernels, on their way
into LDS or registers, and multiplied on v_mfma_f32_*_bf16; tensors stay fp32 in HBM, accumulators, batch-norm statistics,
softmax statistics, LayerNorm, the loss and the optimizer stay fp32 -- the mixed-precision recipe of the reference's
``--amp`` switch (tools/train.py:86-99: AmpOptimWrapper around the same model) with bf16 instead of fp16 operands.

The mode is read when an op runs forward and is remembered for that op's backward.
"""
from __future__ import annotations

import contextlib

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

"""Algorithmic work of the launches of one step (bench.py roofline accounting): per kernel family the number
of launches, the flops and the HBM bytes an ideal implementation would move (every operand read once, every result written
once; gathered rows are NOT multiplied by their reuse).  Off unless bench.py turns it on; the ops then also hand their
flops to the C side, whose HIP events time the launches of each family on the stream they run on."""
from __future__ import annotations

ON = False
ACC = {}


def enable(on: bool):
    global ON
    ON = bool(on)


def reset():
    ACC.clear()


def add(kind: str, flops: float, nbytes: float, launches: int = 1):
    if ON:
        a = ACC.setdefault(kind, [0, 0.0, 0.0])
        a[0] += launches
        a[1] += flops
        a[2] += nbytes


def snapshot():
    return {k: dict(launches=v[0], flops=v[1], bytes=v[2]) for k, v in ACC.items()}

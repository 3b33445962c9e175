"""The arithmetic behind the default fp32 path (unidet3d_amd/csrc/u3d_common.h: split3_pair), restated
in numpy bit operations and checked on the CPU: the split of an fp32 value into three bf16 pieces is EXACT, each piece is a bf16
value, the remainders have the stated bounds, and the six products kept by the kernels reproduce the fp32 product to the stated
error.  (The kernels themselves are compared with fp64 on the GPU: tests/test_gpu_kernels.py, tests/test_gpu_model.py.)"""
import numpy as np


def split3(x: np.ndarray):
    """h = x rounded to 8 significant bits (add half an ulp to the bit pattern, clear the low 16 bits), m = x - h truncated to
    8 bits, l = x - h - m -- the instruction sequence of split3_pair, one value at a time."""
    x = x.astype(np.float32)
    xb = x.view(np.uint32)
    h = ((xb + np.uint32(0x8000)) & np.uint32(0xffff0000)).view(np.float32)
    r1 = (x - h).astype(np.float32)
    m = (r1.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    l = (r1 - m).astype(np.float32)
    return h, m, l


def _is_bf16(v: np.ndarray) -> bool:
    return bool(np.all((v.astype(np.float32).view(np.uint32) & np.uint32(0xffff)) == 0))


def _values(n=200_000, seed=0):
    rng = np.random.default_rng(seed)
    v = (rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 6.0)).astype(np.float32)           # 1e-8 .. 1e8, both signs
    edge = np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0 / 256.0, 1.99999988, 3.0e38, 1.0e-30,
                     0.00390625, 1.00390625, 1.005859375], dtype=np.float32)
    return np.concatenate([v, edge, -edge])


def test_split_is_exact_and_every_piece_is_a_bf16_value():
    x = _values()
    h, m, l = split3(x)
    assert _is_bf16(h) and _is_bf16(m) and _is_bf16(l)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)), x.astype(np.float64))
    nz = x != 0
    ax = np.abs(x[nz].astype(np.float64))
    r1 = np.abs(x[nz].astype(np.float64) - h[nz])
    assert np.all(r1 <= ax * 2.0 ** -8 * (1 + 1e-12))                 # rounded top piece: at most half an 8-bit ulp
    assert np.all(np.abs(l[nz].astype(np.float64)) <= r1 * 2.0 ** -7 + 1e-300)        # truncated middle piece: less than one of its ulps
    # the remainder after the top piece takes either sign (that is what keeps the dropped cross terms unbiased)
    s = np.sign(x[nz].astype(np.float64) - h[nz]) * np.sign(x[nz])
    assert 0.4 < np.mean(s[s != 0] > 0) < 0.6


def test_six_products_reproduce_the_fp32_product():
    """x y - (hh' + hm' + mh' + hl' + mm' + lh') = m l' + l m' + l l': at most 2^-22 |x y|, ~2^-26 on average and zero-mean --
    the level of an fp32 multiply-add's own rounding (2^-24 worst, zero-mean)."""
    rng = np.random.default_rng(1)
    x, y = _values(100_000, 2), _values(100_000, 3)
    n = min(len(x), len(y))
    x, y = x[:n], y[:n]
    keep = (x != 0) & (y != 0) & (np.abs(x.astype(np.float64) * y) < 1e37) & (np.abs(x.astype(np.float64) * y) > 1e-30)
    x, y = x[keep], y[keep]
    hx, mx, lx = [v.astype(np.float64) for v in split3(x)]
    hy, my, ly = [v.astype(np.float64) for v in split3(y)]
    six = hx * hy + (hx * my + mx * hy) + (hx * ly + mx * my + lx * hy)
    exact = x.astype(np.float64) * y.astype(np.float64)
    rel = (six - exact) / exact                                        # signed relative to the product: < 0 = short in magnitude
    assert np.max(np.abs(rel)) <= 2.0 ** -22
    assert np.mean(np.abs(rel)) < 2.0 ** -25
    assert abs(np.mean(rel)) < 2.0 ** -30                              # no bias towards zero or away from it
    # every kept product is exact in fp32 (8-bit x 8-bit significands): the MFMA adds exact terms
    for a, b in ((hx, hy), (hx, my), (mx, hy), (hx, ly), (mx, my), (lx, hy)):
        p = a * b
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)


def test_truncating_the_top_piece_would_bias_the_dropped_terms():
    """The 11-instruction variant (truncate h as well) was measured 2.5x worse end to end: its remainders all
    carry the sign of x, so the dropped terms are biased towards zero and up to 2^-20 of a product."""
    x, y = _values(100_000, 4), _values(100_000, 5)
    n = min(len(x), len(y)); x, y = x[:n], y[:n]
    keep = (x != 0) & (y != 0) & (np.abs(x.astype(np.float64) * y) < 1e37) & (np.abs(x.astype(np.float64) * y) > 1e-30)
    x, y = x[keep], y[keep]

    def trunc3(v):
        h = (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
        r1 = (v - h).astype(np.float32)
        m = (r1.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
        return [t.astype(np.float64) for t in (h, m, (r1 - m).astype(np.float32))]
    hx, mx, lx = trunc3(x); hy, my, ly = trunc3(y)
    six = hx * hy + (hx * my + mx * hy) + (hx * ly + mx * my + lx * hy)
    exact = x.astype(np.float64) * y.astype(np.float64)
    rel = (six - exact) / exact
    assert np.mean(rel) < -2.0 ** -26                                  # systematically short in magnitude
    assert np.max(np.abs(rel)) > 2.0 ** -22

"""The arithmetic claim behind the bf16x3 mode (csrc/mfma_mlp_x3.h), checked on
the CPU with an exact emulation: an fp32 value splits EXACTLY into three bf16
terms (round to nearest even), and the six partial products of order <= 2
reproduce a product to 2^-23 relative.  No GPU needed."""
import numpy as np


def bf16_rn(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    lsb = (b >> 16) & 1
    b = (b + 0x7FFF + lsb) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    x0 = bf16_rn(x)
    r1 = (x - x0).astype(np.float32)          # exact in fp32 (checked below)
    x1 = bf16_rn(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_rn(r2)
    return x0, x1, x2, r1, r2


def _values(n, seed):
    g = np.random.default_rng(seed)
    mant = g.uniform(1.0, 2.0, n)
    expo = g.integers(-20, 20, n)
    sign = g.choice([-1.0, 1.0], n)
    v = (sign * mant * np.exp2(expo)).astype(np.float32)
    edge = np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 255.0 / 256, 3.0e38, 1.0e-30,
                     65504.0, 1.0 - 2.0 ** -24, 0.1, -0.3], dtype=np.float32)
    return np.concatenate([v, edge])


def test_three_bf16_terms_reproduce_an_fp32_value_exactly():
    x = _values(200000, 1)
    x0, x1, x2, r1, r2 = split3(x)
    xd = x.astype(np.float64)
    # the residuals are exact in fp32 and the last one is a bf16 number
    assert np.array_equal(r1.astype(np.float64), xd - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), xd - x0.astype(np.float64) - x1.astype(np.float64))
    assert np.array_equal(x2, r2)
    assert np.array_equal(x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64), xd)
    # term sizes: |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|
    nz = x != 0
    assert np.all(np.abs(x1[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(x2[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_partial_products_are_an_fp32_grade_product():
    x, w = _values(100000, 2), _values(100000, 3)[::-1].copy()
    keep = (np.abs(x) < 1e18) & (np.abs(w) < 1e18) & (np.abs(x) > 1e-18) & (np.abs(w) > 1e-18)
    x, w = x[keep], w[keep]
    xs, ws = split3(x)[:3], split3(w)[:3]
    d = np.float64
    six = (xs[2].astype(d) * ws[0] + xs[1].astype(d) * ws[1] + xs[0].astype(d) * ws[2] +
           xs[1].astype(d) * ws[0] + xs[0].astype(d) * ws[1] + xs[0].astype(d) * ws[0])
    exact = x.astype(d) * w.astype(d)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23, rel.max()
    # every partial product of two bf16 values is exact in fp32
    p = (xs[0] * ws[0]).astype(np.float32)
    assert np.array_equal(p.astype(d), xs[0].astype(d) * ws[0].astype(d))
    # for comparison: one fp16 product (tiny-cuda-nn's operands) is 2^-11-grade
    with np.errstate(over="ignore"):
        h = x.astype(np.float16).astype(d) * w.astype(np.float16).astype(d)
    ok = np.isfinite(h) & (np.abs(x) < 6e4) & (np.abs(w) < 6e4) & (np.abs(x) > 1e-4) & (np.abs(w) > 1e-4)
    assert (np.abs(h - exact) / np.abs(exact))[ok].max() > 2.0 ** -13

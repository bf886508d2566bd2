"""The arithmetic MODEL of the f16x2 operand split (csrc/mfma_mlp_h2.h), in
numpy: the error bounds the header states, checked numerically over twelve
decades -- independent of any GPU.  (The kernels themselves are held to an
fp64 evaluation in tests/test_gpu_parity.py::test_f16x2_*.)"""
import numpy as np

LO = np.float32(2048.0)


def split(x):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * LO).astype(np.float16)   # x - hi is exact in fp32
    return hi, lo


def rebuild(hi, lo):
    return hi.astype(np.float64) + lo.astype(np.float64) / 2048.0


def test_pair_carries_22_bits_down_to_2_pow_minus_13_and_2_pow_minus_36_below():
    rng = np.random.default_rng(0)
    for e in range(-30, 16):
        top = 2.0 if e < 15 else 65504.0 / 32768.0      # the last binade ends at f16's maximum
        x = (rng.uniform(1.0, top, 20000) * 2.0 ** e * rng.choice([-1.0, 1.0], 20000)).astype(np.float32)
        x = np.clip(x, -65504.0, 65504.0)
        err = np.abs(rebuild(*split(x)) - x.astype(np.float64))
        if e >= -13:
            assert (err <= 2.0 ** -23 * np.abs(x) * 1.0000001).all(), e
        assert (err <= np.maximum(2.0 ** -23 * np.abs(x), 2.0 ** -36)).all(), e


def test_unscaled_second_term_would_lose_the_small_values():
    """Why the 2^11 scale is there: with lo = f16(x - hi) a feature of 1e-4
    keeps ~11 bits."""
    x = np.float32(1.2345e-4)
    hi = x.astype(np.float16)
    lo_unscaled = (x - hi.astype(np.float32)).astype(np.float16)
    bad = abs(float(hi) + float(lo_unscaled) - float(x)) / float(x)
    good = abs(float(rebuild(*split(x))) - float(x)) / float(x)
    assert bad > 1e-5 and good <= 2.0 ** -23


def test_three_partial_products_are_fp32_grade_for_a_dot_product():
    """cross = xh*wl + xl*wh (scale 2^11), main = xh*wh, result = main + cross/2^11:
    against the exact dot product, next to a plain fp32 dot product."""
    rng = np.random.default_rng(1)
    K, M = 64, 4000
    x = (rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-3, 1, (M, 1))).astype(np.float32)
    w = (rng.standard_normal(K) * 0.3).astype(np.float32)
    xh, xl = split(x)
    wh, wl = split(w)
    f = lambda a: a.astype(np.float64)          # f16 products are exact in fp32 / fp64
    cross = (f(xh) * f(wl) + f(xl) * f(wh)).sum(-1)
    main = (f(xh) * f(wh)).sum(-1)
    got = main + cross / 2048.0
    exact = (f(x) * f(w)).sum(-1)
    plain32 = (x * w).sum(-1, dtype=np.float32).astype(np.float64)
    scale = (np.abs(f(x)) * np.abs(f(w))).sum(-1)
    e_h2 = np.abs(got - exact) / scale
    e_32 = np.abs(plain32 - exact) / scale
    # operand representation (2 x 2^-23) + the dropped xl*wl term (2^-24)
    assert e_h2.max() <= 3.0 * 2.0 ** -23
    assert np.median(e_h2) <= 2.0 * np.median(e_32) + 2.0 ** -26


def test_hidden_scale_is_exact():
    """First-layer weights x 2^-4, last-layer weights x 2^4 (ucsa_mlp_pack_h2):
    powers of two commute with every fp32 operation of a bias-free ReLU net."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((100, 32)).astype(np.float32)
    w1 = rng.standard_normal((64, 32)).astype(np.float32)
    w2 = rng.standard_normal((16, 64)).astype(np.float32)
    a = np.maximum(x @ w1.T, 0) @ w2.T
    b = np.maximum(x @ (w1 * np.float32(0.0625)).T, 0) @ (w2 * np.float32(16.0)).T
    assert np.array_equal(a, b)

"""The FORMAT of the 8-byte packed bin records of the hash-grid backward
(csrc/hashgrid_bwd.hip p64_pack / p64_unpack), restated in numpy: what a record
can hold and how far a value moves.  (The kernels are held to the 16-byte
records in tests/test_gpu_backward.py::test_packed_bin_records_match_fp32_records.)"""
import numpy as np


def value_bits(L):
    return min(32, (64 - L) // 2)


def rnd(f, V):
    b = np.asarray(f, np.float32).view(np.uint32).astype(np.uint64)
    drop = 32 - V
    if drop:
        r = b + ((1 << (drop - 1)) - 1) + ((b >> drop) & 1)               # nearest even
        # non-finite values are truncated instead: the add would carry an all-ones
        # NaN payload into the sign bit or wrap it around in 32 bits
        b = np.where((b & 0x7F800000) == 0x7F800000, b, r) >> drop
    return b


def pack(loc, vx, vy, L):
    V = value_bits(L)
    return (np.asarray(loc, np.uint64) | (rnd(vx, V) << np.uint64(L))
            | (rnd(vy, V) << np.uint64(L + V)))


def unpack(w, L):
    V = value_bits(L)
    drop = 32 - V
    vmask = np.uint64((1 << V) - 1)
    loc = (w & np.uint64((1 << L) - 1)).astype(np.uint32)
    vx = (((w >> np.uint64(L)) & vmask) << np.uint64(drop)).astype(np.uint32).view(np.float32)
    vy = (((w >> np.uint64(L + V)) & vmask) << np.uint64(drop)).astype(np.uint32).view(np.float32)
    return loc, vx, vy


def test_reference_grid_records_round_to_26_bits():
    """2^19-entry levels in 256 bins: 11 index bits, 26-bit values (sign, 8
    exponent, 17 mantissa bits): relative error <= 2^-18 over fp32's range."""
    L = 11
    assert value_bits(L) == 26 and L + 2 * 26 <= 64
    rng = np.random.default_rng(0)
    n = 200000
    loc = rng.integers(0, 1 << L, n)
    vx = (rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
    vy = (rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
    l2, x2, y2 = unpack(pack(loc, vx, vy, L), L)
    assert np.array_equal(l2, loc.astype(np.uint32))
    for a, b in ((vx, x2), (vy, y2)):
        ok = np.isfinite(a) & (np.abs(a) > 1e-37)         # normal fp32 values
        rel = np.abs(b[ok].astype(np.float64) - a[ok]) / np.abs(a[ok])
        assert rel.max() <= 2.0 ** -18


def test_non_finite_values_stay_non_finite_and_zero_stays_zero():
    L = 11
    v = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 3.4e38], np.float32)
    # a NaN with only low mantissa bits set truncates to an infinity: still non-finite
    low_nan = np.array([0x7F800001], np.uint32).view(np.float32)
    # ... and NaNs with an all-ones payload (either sign) stay NaNs: a 32-bit
    # rounding add would wrap 0xFFFFFFFF around to +0.0
    ones_nan = np.array([0xFFFFFFFF, 0x7FFFFFFF, 0xFFFFFFE0], np.uint32).view(np.float32)
    v = np.concatenate([v, low_nan, ones_nan])
    _, x2, _ = unpack(pack(np.zeros(len(v), np.uint64), v, v, L), L)
    assert np.array_equal(np.isfinite(x2), np.isfinite(v))
    assert x2[3] == 0.0 and x2[4] == 0.0


def test_every_bin_width_fits_the_word():
    for L in range(0, 25):
        V = value_bits(L)
        assert L + 2 * V <= 64 and V >= 20
        loc = np.array([(1 << L) - 1 if L else 0], np.uint64)
        l2, x2, y2 = unpack(pack(loc, np.float32(1.5), np.float32(-2.25), L), L)
        assert int(l2[0]) == int(loc[0]) and x2[0] == 1.5 and y2[0] == -2.25

"""The arithmetic behind csrc/hashgrid_bwd.hip's x-pair records (k_grid_bwd_bin_xpair:
the packed records of the training backward since round 6), in numpy: where the corner at x + 1 of a cell lives
relative to the corner at x, for the index functions of the reference's grid
(tiny-cuda-nn: hashed idx = (x ^ y P1 ^ z P2) & (E - 1) with E = 2^19, dense idx =
(x + y res + z res^2) % E)."""
import numpy as np

P1, P2 = np.uint32(2654435761), np.uint32(805459861)
E = np.uint32(1 << 19)
BIN_SHIFT = 11            # 256 bins of 2048 entries per level


def _cells(res, n, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, res, n, dtype=np.uint32), rng.integers(0, res + 1, n, dtype=np.uint32),
            rng.integers(0, res + 1, n, dtype=np.uint32))


def _hashed(x, y, z):
    return (x ^ (y * P1) ^ (z * P2)) & (E - np.uint32(1))


def test_hashed_x_neighbour_is_one_xor_and_shares_the_bin():
    for res in (112, 400, 1023, 2048, 4096):
        x, y, z = _cells(res, 2_000_000, res)
        i0, i1 = _hashed(x, y, z), _hashed(x + np.uint32(1), y, z)
        flip = (x ^ (x + np.uint32(1))) & (E - np.uint32(1))
        assert np.array_equal(i1, i0 ^ flip)                       # what the kernel computes
        same = (i0 >> BIN_SHIFT) == (i1 >> BIN_SHIFT)
        # the pair leaves the bin exactly when x ends in >= 11 one-bits
        trailing_ones = (x & np.uint32(0x7FF)) == np.uint32(0x7FF)
        assert np.array_equal(~same, trailing_ones)
        if res <= 2047:
            assert same.all()
        else:
            assert abs((~same).mean() - 2.0 ** -11) < 2e-4
        # the record stores loc0 and loc0 ^ loc1 (both inside the bin): loc1 comes back
        l0, l1 = i0 & np.uint32(2047), i1 & np.uint32(2047)
        assert np.array_equal((l0 ^ (l0 ^ l1))[same], l1[same])
        # y / z neighbours do NOT share a bin (why there is no "cell record")
        assert ((i0 >> BIN_SHIFT) == (_hashed(x, y + np.uint32(1), z) >> BIN_SHIFT)).mean() < 0.01


def test_dense_x_neighbour_is_the_next_entry():
    res = 64                                       # res^3 = 262144 < 2^19: a dense level
    entries = np.uint32(((res + 1) ** 3 + 7) // 8 * 8)
    x, y, z = _cells(res, 500_000, 5)
    r = np.uint32(res + 1)
    i0 = (x + y * r + z * r * r) % entries
    i1 = (x + np.uint32(1) + y * r + z * r * r) % entries
    nxt = np.where(i0 + np.uint32(1) == entries, np.uint32(0), i0 + np.uint32(1))
    assert np.array_equal(i1, nxt)

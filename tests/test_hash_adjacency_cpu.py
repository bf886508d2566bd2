"""The index property the wide gathers of the hash-grid encoder rely on
(csrc/hashgrid_common.h): with tiny-cuda-nn's hash x ^ (y * P2) ^ (z * P3) the
x-neighbours of an aligned group of 2 (4) cells land in one aligned group of
2 (4) table entries -- one 16-byte access for 8-byte (4-byte) entries.  CPU."""
import numpy as np

P2, P3 = np.uint32(2654435761), np.uint32(805459861)


def _cases(n, seed):
    g = np.random.default_rng(seed)
    gx = g.integers(0, 8192, n, dtype=np.uint32)
    gy = g.integers(0, 8192, n, dtype=np.uint32)
    gz = g.integers(0, 8192, n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = (gy * P2) ^ (gz * P3)
    return gx, h


def test_x_neighbours_share_an_aligned_pair_or_quad_of_entries():
    mask = np.uint32((1 << 19) - 1)
    gx, h = _cases(500000, 5)
    i0 = (gx ^ h) & mask
    i1 = ((gx + np.uint32(1)) ^ h) & mask
    even = (gx & 1) == 0
    # fp32 table (8-byte entries): the pair (x0, x0 + 1) is idx, idx ^ 1 for even x0
    assert np.array_equal(i1[even], i0[even] ^ 1)
    assert np.all((i0[even] >> 1) == (i1[even] >> 1))
    assert np.all((i0[~even] >> 1) != (i1[~even] >> 1))
    # fp16 table (4-byte entries): one aligned quad unless x0 = 3 (mod 4)
    inside = (gx & 3) != 3
    assert np.all((i0[inside] >> 2) == (i1[inside] >> 2))
    assert np.all((i0[~inside] >> 2) != (i1[~inside] >> 2))
    # expected accesses per (y, z) corner pair: 1.5 and 1.25 -> 6 and 5 per sample
    assert abs((1 + (~even).mean()) * 4 - 6.0) < 0.02
    assert abs((1 + (~inside).mean()) * 4 - 5.0) < 0.02

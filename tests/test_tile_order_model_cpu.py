"""The layout of the depth-ordered sample arrays (csrc/hashgrid_sorted.h
``tile_geom``), restated in Python: every 8x8 pixel tile of an image of whole
rows owns a contiguous run of N*T positions, the runs tile [0, N*T) in row-major
tile order whatever the raggedness of the last column / band, and the exact
integer division the sort kernel does in fp32 (``p = (e + 0.5) * (1 / T)``) is
exact over its whole domain.  (The kernels are held to this layout by
tests/test_gpu_parity.py::test_depth_ordered_density_is_bit_identical.)"""
import numpy as np
import pytest


def tile_geom(tile, rows, W, T):
    tiles_x = (W + 7) // 8
    tx, ty = tile % tiles_x, tile // tiles_x
    px0, py0 = tx * 8, ty * 8
    wt, ht = min(8, W - px0), min(8, rows - py0)
    return dict(base=T * (py0 * W + px0 * ht), count=wt * ht * T, wt=wt, ht=ht, px0=px0, py0=py0)


@pytest.mark.parametrize("rows,W,T", [(8, 8, 1), (17, 23, 8), (24, 40, 16), (96, 640, 96),
                                      (240, 320, 256), (3, 5, 7), (9, 641, 33)])
def test_tiles_partition_the_sample_range(rows, W, T):
    tiles = ((W + 7) // 8) * ((rows + 7) // 8)
    runs = sorted((g["base"], g["base"] + g["count"]) for g in
                  (tile_geom(t, rows, W, T) for t in range(tiles)))
    assert runs[0][0] == 0 and runs[-1][1] == rows * W * T
    for (a0, a1), (b0, b1) in zip(runs, runs[1:]):
        assert a1 == b0 and a1 > a0
    # row-major tile order is position order
    bases = [tile_geom(t, rows, W, T)["base"] for t in range(tiles)]
    assert bases == sorted(bases)
    # every ray lies in exactly one tile, at a valid local pixel
    seen = np.zeros(rows * W, np.int32)
    for t in range(tiles):
        g = tile_geom(t, rows, W, T)
        for ly in range(g["ht"]):
            for lx in range(g["wt"]):
                seen[(g["py0"] + ly) * W + g["px0"] + lx] += 1
    assert (seen == 1).all()


def test_fp32_pixel_index_of_an_element_is_exact():
    """e = p * T + s, e < 64 T <= 65536: (float(e) + 0.5f) * (1.0f / T) truncated is p."""
    for T in list(range(1, 200)) + [255, 256, 257, 511, 512, 1000, 1023, 1024]:
        e = np.arange(64 * T, dtype=np.uint32)
        inv_T = np.float32(1.0) / np.float32(T)
        p = ((e.astype(np.float32) + np.float32(0.5)) * inv_T).astype(np.uint32)
        assert np.array_equal(p, e // T), T


def test_byte_counters_of_the_second_sort_cannot_overflow():
    """Four pixel counters to a 32-bit word: a run holds 64 samples, so a count
    and its exclusive prefix are <= 64 < 256 -- no carry into the neighbour."""
    rng = np.random.default_rng(0)
    for _ in range(200):
        pix = rng.integers(0, 64, 64)                    # one run of 64 ranks
        if rng.random() < 0.2:
            pix[:] = rng.integers(0, 64)                 # all from one pixel
        words = np.zeros(16, np.uint64)
        for p in pix:
            words[p >> 2] += np.uint64(1) << np.uint64(8 * (p & 3))
        assert (words < (1 << 32)).all()
        counts = np.array([(int(words[p >> 2]) >> (8 * (p & 3))) & 255 for p in range(64)])
        assert np.array_equal(counts, np.bincount(pix, minlength=64))
        prefix = np.concatenate([[0], np.cumsum(counts)[:-1]])
        assert prefix.max() <= 64 and (prefix + counts).max() <= 64

"""csrc/hashgrid_bwd.hip's run_plan / run_sum in numpy: the segmented wave
scan on the DPP ladder (row_shr 1 / 2 / 4 / 8 inside the four 16-lane rows,
row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3) must leave the SUM OF
ITS RUN in the last lane of every run, like the shuffle ladder (__shfl_up by 1, 2,
4, 8, 16, 32) it replaces -- for any pattern of run heads.  Lanes without a DPP
source keep the identity, as `update_dpp(old = identity, ..., bound_ctrl = false)`
does; a step adds only where the lane's own flag is still clear."""
import numpy as np

LANES = np.arange(64)


def _src_row_shr(d):
    src = LANES - d
    return np.where((LANES & 15) >= d, src, -1)


def _src_bcast15():
    row = LANES >> 4
    return np.where(row & 1, (row - 1) * 16 + 15, -1)        # rows 1, 3 <- last lane of the row before


def _src_bcast31():
    return np.where(LANES >= 32, 31, -1)                     # rows 2, 3 <- lane 31


DPP_STEPS = [_src_row_shr(1), _src_row_shr(2), _src_row_shr(4), _src_row_shr(8), _src_bcast15(), _src_bcast31()]
SHFL_STEPS = [np.where(LANES >= d, LANES - d, -1) for d in (1, 2, 4, 8, 16, 32)]


def _scan(head, v, steps):
    """run_plan + run_sum: flags first (add[s] per lane), then the values."""
    f = head.astype(np.int64).copy()
    adds = []
    for src in steps:
        has = src >= 0
        of = np.where(has, f[np.maximum(src, 0)], 1)         # identity 1 for the flags
        add = has & (f == 0)
        f = np.where(add, f | of, f)
        adds.append(add)
    v = v.astype(np.float64).copy()
    for src, add in zip(steps, adds):
        o = np.where(src >= 0, v[np.maximum(src, 0)], 0.0)   # identity 0 for the values
        v = np.where(add, v + o, v)
    return v


def test_the_dpp_ladder_sums_every_run_like_the_shuffle_ladder():
    rng = np.random.default_rng(0)
    for trial in range(3000):
        p = rng.choice([0.02, 0.1, 0.3, 0.7])
        head = rng.random(64) < p
        head[0] = True
        if trial == 0:
            head[1:] = False                                 # one run over the whole wave
        if trial == 1:
            head[:] = True                                   # 64 runs of one
        v = rng.integers(-1000, 1000, 64).astype(np.float64)  # integers: sums exact in any order
        run_id = np.cumsum(head) - 1
        want = np.zeros(64)
        np.add.at(want, run_id, v)
        tail = np.append(head[1:], True)                     # last lane of each run
        for name, steps in (("dpp", DPP_STEPS), ("shuffle", SHFL_STEPS)):
            got = _scan(head, v, steps)
            assert np.array_equal(got[tail], want[run_id[tail]]), (name, trial)
            # and every lane holds the sum of its run up to itself (inclusive scan)
            incl = np.array([v[np.flatnonzero(run_id == run_id[i])[0]:i + 1].sum() for i in range(64)])
            assert np.array_equal(got, incl), (name, trial)

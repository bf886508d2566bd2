"""tests/golden/g9_trajectory_long.npz (the oracle trainer's side of the long-horizon
quality test, tests/test_gpu_trajectory.py) is what tests/golden/make_trajectory_golden.py
produces from the committed oracle: its layout, the statistics the GPU test compares
against, and -- by re-running the first steps of the oracle trainer here -- its losses.
The same for the short horizon's two runs (g9_trajectory_short.npz)."""
import numpy as np
import torch

from tests import test_gpu_trajectory as tt
from tests.util import load_golden


def test_fixture_layout_and_statistics():
    g = load_golden("g9_trajectory_long.npz")
    ck = tuple(int(x) for x in g["checkpoints"])
    assert ck == tt.LONG.checkpoints and int(g["steps"]) == tt.LONG.steps and int(g["rays"]) == tt.LONG.n
    psnr, miou, losses = g["psnr"].numpy(), g["miou"].numpy(), g["losses"].numpy()
    runs = psnr.shape[0]
    assert runs == 6 and psnr.shape == miou.shape == (runs, len(ck)) and losses.shape == (runs, tt.LONG.steps)
    assert sorted(int(t) for t in g["threads"]) == [2, 3, 4, 5, 7, 8]          # six different round-offs
    late = np.array(ck) >= 424
    st = np.array([tt.long_run_stats(p, m, late) for p, m in zip(psnr, miou)])
    # what the runs agree on (and what the GPU test is held to): mean PSNR within ~1 dB of
    # each other, converged-window median mIoU within half a point
    assert 35.5 < st[:, 0].min() and st[:, 0].max() < 37.5 and np.ptp(st[:, 0]) < 1.0
    assert np.ptp(st[:, 1]) <= 0.5 and 74.0 < st[:, 1].mean() < 75.5
    # ... and what they do not: single checkpoints, and a run's late-window MEAN mIoU
    assert np.abs(psnr - psnr.mean(0)).max() > 2.0
    assert np.ptp(miou[:, late].mean(1)) > 2.0
    # every run learned: loss falls by two orders of magnitude, the head converges
    assert (losses[:, -20:].mean(1) < 0.02 * losses[:, 0]).all()
    assert (miou[:, :40].mean(1) < 45.0).all() and (np.median(miou[:, late], 1) > 70.0).all()
    # the runs share their start (same init, same draws) and decorrelate later
    assert np.ptp(losses[:, :5], 0).max() <= 1e-4 * losses[0, 0]
    assert np.ptp(losses[:, 300:], 0).max() > 1e-4


def test_the_first_steps_of_the_oracle_trainer_reproduce_the_fixture():
    g = load_golden("g9_trajectory_long.npz")
    frames = tt._frames()
    draws, u_eval = tt._draws(tt.LONG)
    torch.manual_seed(0)
    _, losses, _ = tt._train_oracle(frames, draws[:3], u_eval, False, checkpoints=(), raw_quals=True)
    want = g["losses"].numpy()[:, :3]
    for k in range(3):
        assert abs(losses[k] - want[:, k].mean()) <= 2e-5 * max(1.0, abs(want[:, k].mean())), (k, losses[k], want[:, k])


def test_short_fixture_layout_and_what_the_runs_show():
    for kind in ("fp32", "tcnn"):
        q, losses, _ = tt.short_oracle(kind)
        assert len(losses) == tt.SHORT.steps and len(q["per_checkpoint_train_psnr"]) == len(tt.SHORT.checkpoints)
        losses = np.array(losses)
        assert losses[-10:].mean() < 0.6 * losses[0] and q["train"][0] > 14.0        # tt._compare's premises
        assert 0.0 < q["train"][1] <= 100.0 and 0.0 < q["held"][1] <= 100.0
    a, b = np.array(tt.short_oracle("fp32")[1]), np.array(tt.short_oracle("tcnn")[1])
    assert abs(a[0] - b[0]) < 1e-2 * a[0] and (a != b).any()                          # same start, other numerics


def test_the_first_steps_of_the_short_oracle_runs_reproduce_the_fixture():
    frames = tt._frames()
    draws, u_eval = tt._draws(tt.SHORT)
    for kind in ("fp32", "tcnn"):
        torch.manual_seed(0)
        _, losses, _ = tt._train_oracle(frames, draws[:3], u_eval, kind == "tcnn", checkpoints=(), raw_quals=True)
        want = tt.short_oracle(kind)[1][:3]
        for k in range(3):
            assert abs(losses[k] - want[k]) <= 2e-5 * max(1.0, abs(want[k])), (kind, k, losses[k], want[k])

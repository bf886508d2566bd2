"""Generates tests/golden/g9_trajectory_long.npz: the CPU oracle trainer's side of
tests/test_gpu_trajectory.py::test_long_horizon_quality_* -- 600 steps of 4096 rays x
(16+16) samples with the reference's losses and Adam settings (reference
nr4seg/lightning/joint_train_lightning_net.py:473-513, :897-919) on the synthetic room,
quality (per-view-mean PSNR and mIoU on the 8 training views, as the reference's test
loop measures them: nr4seg/utils/metrics.py:13-65, joint_train_lightning_net.py:648-693)
after every 5th step from step 100.  TWO runs with different BLAS thread counts (another
summation order: the oracle's own run-to-run spread is part of the fixture).

    python tests/golden/make_trajectory_golden.py [threads ...]     (~30 min per pair on 8 cores)

The committed fixture holds SIX runs (thread counts 4, 3, 5, 2, 7, 8: `... 4 3`, then
`APPEND=1 ... 5 2`, then `APPEND=1 PAR=1 ... 7 8`; APPEND adds runs to the existing file).

    python tests/golden/make_trajectory_golden.py short              (~5 min)

writes tests/golden/g9_trajectory_short.npz instead: the two oracle runs of the SHORT
horizon (150 steps x 2048 rays; fp32, and the fp16-emulating forward of tiny-cuda-nn's
numerics) that test_trajectory_quality_* compare the HIP runs with -- until round 6 they
were trained inside the GPU suite (147 s of its 500).

No GPU, no reference code: oracle/ + the synthetic scene's analytic ray casting only."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(args):
    tag, threads = args
    import torch
    torch.set_num_threads(threads)
    from tests import test_gpu_trajectory as tt
    frames = tt._frames()
    draws, u_eval = tt._draws(tt.LONG)
    t0 = time.time()
    log = open(os.path.join(ROOT, "tests", "golden", f"_trajectory_{tag}{threads}.log"), "w")

    def progress(k, q, losses):
        log.write(f"step {k + 1} t {time.time() - t0:.0f}s psnr {q['train'][0]:.2f} miou {q['train'][1]:.1f} "
                  f"held {q['held'][0]:.2f} loss {np.mean(losses[-20:]):.5f}\n")
        log.flush()

    quals, losses, _ = tt._train_oracle(frames, draws, u_eval, False, tt.LONG.checkpoints,
                                        raw_quals=True, progress=progress)
    return dict(psnr=[q["train"][0] for q in quals], miou=[q["train"][1] for q in quals],
                held_psnr=[q["held"][0] for q in quals], held_miou=[q["held"][1] for q in quals],
                losses=losses, threads=threads, seconds=time.time() - t0)


def short_worker(args):
    kind, threads = args
    import torch
    torch.set_num_threads(threads)
    from tests import test_gpu_trajectory as tt
    frames = tt._frames()
    draws, u_eval = tt._draws(tt.SHORT)
    t0 = time.time()
    quals, losses, _ = tt._train_oracle(frames, draws, u_eval, kind == "tcnn", tt.SHORT.checkpoints,
                                        raw_quals=True)
    return kind, dict(psnr=[q["train"][0] for q in quals], miou=[q["train"][1] for q in quals],
                      held_psnr=[q["held"][0] for q in quals], held_miou=[q["held"][1] for q in quals],
                      losses=losses, seconds=time.time() - t0)


def main_short():
    import torch
    from tests import test_gpu_trajectory as tt
    threads = max(1, (os.cpu_count() or 2) // 2)
    with mp.get_context("spawn").Pool(2) as pool:
        runs = dict(pool.map(short_worker, [("fp32", threads), ("tcnn", threads)]))
    out = os.path.join(ROOT, "tests", "golden", "g9_trajectory_short.npz")
    arrays = {f"{kind}_{k}": np.array(v, dtype=np.float64) for kind, r in runs.items() for k, v in r.items()}
    np.savez_compressed(
        out, checkpoints=np.array(tt.SHORT.checkpoints), steps=np.array(tt.SHORT.steps),
        rays=np.array(tt.SHORT.n), seed=np.array(tt.SHORT.seed), threads=np.array(threads), **arrays)
    print("wrote", out, "torch", torch.__version__, {k: round(r["seconds"]) for k, r in runs.items()})


def main():
    import torch
    from tests import test_gpu_trajectory as tt
    if sys.argv[1:2] == ["short"]:
        return main_short()
    threads = [int(x) for x in sys.argv[1:]] or [5, 3]
    # PAR runs at a time: keep the sum of their thread counts at or below the core count
    # (oversubscribed OpenMP teams spin: two runs with 7 + 6 threads on 8 cores did not
    # reach step 100 in half an hour)
    with mp.get_context("spawn").Pool(int(os.environ.get("PAR", "2"))) as pool:
        runs = pool.map(worker, [("R", t) for t in threads])
    out = os.path.join(ROOT, "tests", "golden", "g9_trajectory_long.npz")
    new = {k: np.array([r[k] for r in runs], dtype=np.float64)
           for k in ("psnr", "miou", "held_psnr", "held_miou", "losses")}
    new["threads"] = np.array([r["threads"] for r in runs])
    new["seconds"] = np.array([r["seconds"] for r in runs])
    if os.environ.get("APPEND") == "1" and os.path.exists(out):
        old = np.load(out)
        assert tuple(old["checkpoints"]) == tuple(tt.LONG.checkpoints)
        new = {k: np.concatenate([old[k], v], 0) for k, v in new.items()}
    np.savez_compressed(
        out, checkpoints=np.array(tt.LONG.checkpoints), steps=np.array(tt.LONG.steps),
        rays=np.array(tt.LONG.n), seed=np.array(tt.LONG.seed), **new)
    print("wrote", out, "torch", torch.__version__)


if __name__ == "__main__":
    main()

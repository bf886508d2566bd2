"""Exploration behind tests/test_gpu_trajectory.py's LONG run (round 6): the oracle and
the HIP path trained side by side for 600 steps of 4096 rays with a checkpoint every 25
steps from step 100, per-checkpoint PSNR / mIoU of both dumped as JSON -- to choose the
horizon and the window on which the +-0.5 dB / +-0.5 pt bound is asserted.
    python tests/scripts/trajectory_long_explore.py gpurun_out/r6/trajectory_long.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tests import test_gpu_trajectory as tt


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "trajectory_long.json"
    import multiprocessing as mp
    from tests.conftest import _effective_cores
    run = tt.LONG._replace(checkpoints=tuple(range(99, tt.LONG.steps, 25)))
    frames = tt._frames()
    draws, u_eval = tt._draws(run)
    threads = max(1, _effective_cores() - 2)
    pool = mp.get_context("spawn").Pool(1)
    t0 = time.time()
    job = pool.apply_async(tt._oracle_worker, ((frames, draws, u_eval, False, threads, run.checkpoints),))
    res = {"checkpoints": list(run.checkpoints), "rays": run.n, "steps": run.steps}
    for prec in ("bf16x3", "fp32"):
        for rep in range(2):
            t1 = time.time()
            q, losses, info = tt._train_hip(frames, draws, u_eval, prec, checkpoints=run.checkpoints)
            res[f"hip_{prec}_{rep}"] = {"psnr": q["per_checkpoint_train_psnr"], "miou": q["per_checkpoint_train_miou"],
                                        "held": q["held"], "loss_last20": sum(losses[-20:]) / 20, "s": time.time() - t1}
            print(prec, rep, res[f"hip_{prec}_{rep}"], flush=True)
    q, losses, _ = job.get(timeout=3000)
    res["oracle"] = {"psnr": q["per_checkpoint_train_psnr"], "miou": q["per_checkpoint_train_miou"],
                     "held": q["held"], "loss_last20": sum(losses[-20:]) / 20, "s": time.time() - t0, "threads": threads}
    print("oracle", res["oracle"], flush=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()

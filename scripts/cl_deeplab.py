#!/usr/bin/env python3
"""Drop-in for reference ``scripts/cl_deeplab.py`` (BASELINE cfg5, the
continual loop ``run_scripts/multi_step.sh`` launches): the ten scenes of
``SCENE_ORDER`` (:11-22) one stage at a time through ``train_joint.train``,
stage ``i`` named ``<exp_name>/stage_<i>`` (:68-70), the segmentation network
of stage ``i`` initialised from stage ``i-1``'s ``deeplab.ckpt`` (:77-82) and
only stage 0 from the pre-training checkpoint with the Lightning-key rewrite
(``load_pretrain``, :74-76); ``exp["scenes"]`` grows by one scene per stage
(:66), which is what makes the joint loader replay the earlier scenes
(``cl.replay_buffer_size``).

What differs:
  * ScanNet is absent: when ``env["scannet"]/<scene>/transforms_train.json``
    does not exist and the experiment has a ``synthetic:`` block, the ten
    synthetic rooms (seeds 0-9) are first written there in the
    reference's on-disk layout, so every stage runs through the
    ``ScanNetNGPJoint`` mirror: predict pass -> nerf_image / nerf_label PNGs ->
    next stage's replay.
  * the previous stage's checkpoint is looked up under ``env["results"]``
    (the reference hard-codes "experiments", its default ``results``).
  * stage 0 without a pre-training checkpoint (``general.checkpoint_load``
    empty or missing -- there is no network to fetch one) starts from the
    random initialisation, with a warning.
  * no WandB (out of scope); ``--scenes N`` (extra) limits the loop to the
    first N scenes, ``--limit_batches`` caps every loop (smoke runs).
Under torchrun every rank runs the loop; rank 0 alone writes files.
"""
import argparse
import copy
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ucsa_neural_rendering_amd import ROOT_DIR  # noqa: E402
from ucsa_neural_rendering_amd.utils import load_yaml  # noqa: E402
from scripts.train_joint import train  # noqa: E402

SCENE_ORDER = [f"scene{i:04d}_00" for i in range(10)]  # reference :11-22


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--exp", default="cfg/exp/multi_step/cl_base.yml")
    p.add_argument("--exp_name", default="debug")
    p.add_argument("--seed", default=123, type=int)
    p.add_argument("--fix_nerf", action="store_true")
    p.add_argument("--project_name", default="test_one_by_one")
    p.add_argument("--nerf_train_epoch", default=10, type=int)
    p.add_argument("--joint_train_epoch", default=10, type=int)
    p.add_argument("--scenes", default=len(SCENE_ORDER), type=int,
                   help="extra: only the first N scenes of SCENE_ORDER")
    p.add_argument("--limit_batches", default=None, type=int,
                   help="extra: cap batches per loop (smoke runs)")
    return p.parse_args(argv)


def ensure_synthetic_scenes(exp, env, scenes):
    """Write the synthetic rooms in the ScanNet layout where a scene is
    missing (rank 0; the other ranks wait at the barrier)."""
    syn = exp.get("synthetic")
    root = env["scannet"]
    missing = [s for s in scenes if not os.path.exists(
        os.path.join(root, s, "transforms_train.json"))]
    if not missing:
        return
    if syn is None:
        raise FileNotFoundError(
            f"{missing} not under {root} and the experiment has no "
            "`synthetic:` block to generate them from")
    import torch
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd.dataset.synthetic_export import export
    rank, _, world = udist.init_from_env()
    if rank == 0:
        for s in missing:
            seed = int(s[5:9])
            export(root, seed, int(syn.get("n_views", 20)),
                   int(syn.get("H", 240)), int(syn.get("W", 320)),
                   device="cuda" if torch.cuda.is_available() else "cpu",
                   scene_name=s, palette_seed=syn.get("palette_seed"))
    if udist.active():
        torch.distributed.barrier()


def stage_plan(exp, exp_name, results_dir, n_scenes=len(SCENE_ORDER)):
    """The per-stage settings of reference :62-86 as a list of dicts (pure:
    what ``main`` applies to ``exp`` before each ``train`` call)."""
    plan = []
    prev_stage = "init"
    scenes = []
    for i, new_scene in enumerate(SCENE_ORDER[:n_scenes]):
        scenes.append(new_scene)
        stage = f"stage_{i}"
        if i == 0:
            load_pretrain = True
            old_model_path = exp["general"].get("checkpoint_load", "")
        else:
            load_pretrain = False
            old_model_path = os.path.join(results_dir, exp_name, prev_stage,
                                          "deeplab.ckpt")
        plan.append(dict(scenes=list(scenes), name=f"{exp_name}/{stage}",
                         load_pretrain=load_pretrain,
                         checkpoint_load=old_model_path,
                         resume_from_checkpoint=False,
                         load_from_checkpoint=True))
        prev_stage = stage
    return plan


def main(argv=None, exp=None, env=None):
    args = parse_args(argv)
    env_cfg_path = os.path.join(ROOT_DIR, "cfg/env",
                                os.environ["ENV_WORKSTATION_NAME"] + ".yml")
    exp_cfg_path = os.path.join(ROOT_DIR, args.exp)
    if env is None:
        env = load_yaml(env_cfg_path)
        os.chdir(ROOT_DIR)
    if exp is None:
        exp = load_yaml(exp_cfg_path)
    exp_name = args.exp_name
    exp["exp_name"] = exp_name
    plan = stage_plan(exp, exp_name, env["results"], args.scenes)
    ensure_synthetic_scenes(exp, env, plan[-1]["scenes"])
    results = []
    for st in plan:
        cfg = copy.deepcopy(exp)      # train() rewrites general.name in place
        cfg["scenes"] = st["scenes"]
        cfg["general"]["name"] = st["name"]
        cfg["trainer"]["resume_from_checkpoint"] = st["resume_from_checkpoint"]
        cfg["trainer"]["load_from_checkpoint"] = st["load_from_checkpoint"]
        cfg["general"]["load_pretrain"] = st["load_pretrain"]
        cfg["general"]["checkpoint_load"] = st["checkpoint_load"]
        if st["load_pretrain"] and not (st["checkpoint_load"] and
                                        os.path.exists(st["checkpoint_load"])):
            warnings.warn("stage 0: no pre-training checkpoint at "
                          f"{st['checkpoint_load']!r}; DeepLab starts from its "
                          "random initialisation")
            cfg["general"]["checkpoint_load"] = ""
        print(f"training on: {st['scenes'][-1]}")
        results.append(train(cfg, env, exp_cfg_path, env_cfg_path, args))
    return results


if __name__ == "__main__":
    print(main())

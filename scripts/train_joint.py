#!/usr/bin/env python3
"""Drop-in for reference ``scripts/train_joint.py``: same CLI
(:16-44), same call order (:146-186): NeRF-only fit -> test -> validate ->
joint fit -> test -> predict -> save ``deeplab.ckpt``.  PyTorch-Lightning's
Trainer is replaced by the thin one in ucsa_neural_rendering_amd.lightning
(PL is not installed on the MI355X image); data come from the synthetic scene
data module.  Under torchrun (one process per GPU) the NeRF and DeepLab
gradients are averaged over RCCL inside the module, every rank
trains on its own frames / pixels and evaluates its own shard of the frames
(metrics are reduced; files are written by the rank that owns the frame).
"""
import argparse
import os
import shutil
import sys
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ucsa_neural_rendering_amd import ROOT_DIR, dist as udist  # noqa: E402
from ucsa_neural_rendering_amd.lightning import (JointTrainDataModule,  # noqa: E402
                                                 JointTrainLightningNet,
                                                 Trainer, seed_everything)
from ucsa_neural_rendering_amd.utils import load_yaml  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--exp", default="cfg/exp/synthetic/s00.yml")
    p.add_argument("--exp_name", default="debug")
    p.add_argument("--fix_nerf", action="store_true")
    p.add_argument("--seed", default=123, type=int)
    p.add_argument("--project_name", default="test_one_by_one")
    p.add_argument("--nerf_train_epoch", default=10, type=int)
    p.add_argument("--joint_train_epoch", default=10, type=int)
    p.add_argument("--limit_batches", default=None, type=int,
                   help="extra: cap batches per loop (smoke runs)")
    return p.parse_args(argv)


def train(exp, env, exp_cfg_path, env_cfg_path, args):
    seed_everything(args.seed)
    exp["exp_name"] = args.exp_name
    exp["fix_nerf"] = args.fix_nerf
    rank, local_rank, world = udist.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    model_path = os.path.join(env["results"], exp["general"]["name"])
    if rank == 0:
        if exp["general"]["clean_up_folder_if_exists"]:
            shutil.rmtree(model_path, ignore_errors=True)
        Path(model_path).mkdir(parents=True, exist_ok=True)
        shutil.copy(exp_cfg_path, model_path)
        shutil.copy(env_cfg_path, model_path)
    exp["general"]["name"] = model_path

    # The reference sets cudnn.benchmark (MIOpen's exhaustive solver search for
    # the static-shape DeepLab convolutions).  Here the search result ships with
    # the package (ucsa_neural_rendering_amd/miopen_db), so the default is a
    # look-up; `trainer: {cudnn_benchmark: true|false}` decides otherwise.
    from ucsa_neural_rendering_amd._miopen_db import default_cudnn_benchmark
    bench_before = torch.backends.cudnn.benchmark
    want = exp["trainer"].get("cudnn_benchmark")
    torch.backends.cudnn.benchmark = default_cudnn_benchmark() if want is None else bool(want)
    try:
        return _train(exp, env, args, rank, local_rank, world, model_path)
    finally:  # a process-wide flag: hand it back as found
        torch.backends.cudnn.benchmark = bench_before


def _train(exp, env, args, rank, local_rank, world, model_path):
    model = JointTrainLightningNet(exp, env)
    if udist.active():
        # identical initial parameters on every rank (same seed above), then
        # rank-specific random streams: each rank draws its own pixels,
        # stratified-sampling noise and augmentations (DDP semantics; the
        # gradients are averaged inside the module)
        seed_everything(args.seed + rank)
        udist.broadcast_parameters_(model.to(
            f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu"))
        torch.distributed.barrier()  # rank 0 has created the folder
    exp["seed"] = args.seed          # shared shuffling seed of the samplers
    datamodule = JointTrainDataModule(exp, env)
    datamodule.setup()
    if udist.active():
        torch.distributed.barrier()  # files written by rank 0 during setup
    # ScanNet-layout root: the predict pass writes the PNGs the next stage's
    # replay reads (reference :695-782)
    model.predict_to_disk = bool(getattr(datamodule, "_scannet", False))

    if exp["trainer"].get("load_from_checkpoint") and exp["general"].get(
            "checkpoint_load"):
        ck = torch.load(exp["general"]["checkpoint_load"], map_location="cpu")
        ck = ck["state_dict"]
        if exp["general"].get("load_pretrain", True):
            # reference :116-128: drop aux head, strip the first key component
            ck = {k.split(".", 1)[1]: v for k, v in ck.items()
                  if not k.startswith("_model._model.aux_classifier")}
        model.seg_model.load_state_dict(ck, strict=True)

    kw = dict(default_root_dir=model_path if rank == 0 else None,
              limit_batches=args.limit_batches,
              # `trainer: {prefetch: N}`: batches produced by a background thread
              # (the reference's DataLoader workers; lightning/trainer.py)
              prefetch=int(exp["trainer"].get("prefetch", 0) or 0),
              device=f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu")
    trainer_nerf = Trainer(max_epochs=args.nerf_train_epoch, **kw)
    trainer_joint = Trainer(max_epochs=args.joint_train_epoch,
                            check_val_every_n_epoch=10, **kw)
    results = {}
    model.joint_train = False
    trainer_nerf.fit(model, train_dataloaders=datamodule.train_dataloader_nerf())
    results["test_after_nerf"] = trainer_joint.test(
        model, dataloaders=datamodule.test_dataloader_nerf())
    results["val"] = trainer_joint.validate(model,
                                            dataloaders=datamodule.val_dataloader())
    model.joint_train = True
    trainer_joint.fit(model, train_dataloaders=datamodule.train_dataloader_joint(),
                      val_dataloaders=datamodule.val_dataloader())
    results["test_after_joint"] = trainer_joint.test(
        model, dataloaders=datamodule.train_dataloader_nerf())
    trainer_joint.predict(model, dataloaders=datamodule.predict_dataloader())
    if rank == 0:
        torch.save({"state_dict": model.seg_model.state_dict()},
                   os.path.join(model_path, "deeplab.ckpt"))
    return results


if __name__ == "__main__":
    os.chdir(ROOT_DIR)
    args = parse_args()
    exp_cfg_path = os.path.join(ROOT_DIR, args.exp)
    exp = load_yaml(exp_cfg_path)
    exp["general"]["load_pretrain"] = True
    env_cfg_path = os.path.join(ROOT_DIR, "cfg/env",
                                os.environ["ENV_WORKSTATION_NAME"] + ".yml")
    env = load_yaml(env_cfg_path)
    print(train(exp, env, exp_cfg_path, env_cfg_path, args))

"""`bench.py --mode cfg3`: the joint training step."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import *  # noqa: F401,F403
from .common import masked_fraction
from .seg import conv_flops


def main_cfg3(args, dev, dist, world, rank, backend):
    """--mode cfg3: the joint training step of the LightningModule mirror
    (reference training_step_joint, joint_train_lightning_net.py:363-471) at
    BASELINE cfg3's batch: 8 new-scene frames of 320x240 per rank and step --
    per frame one full no-grad render (256+256 samples, the reference's
    behaviour: it feeds the augmentation / pseudo-label path) and one
    4096-ray NeRF training step (fwd + bwd + Adam), then DeepLabV3 forward /
    backward / Adam on the 8 augmented renders.  `value` = NeRF rays per
    second through the step (rendered + trained), whole job."""
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd.lightning import (JointTrainDataModule,
                                                     JointTrainLightningNet, Trainer)
    import tempfile
    B, Hh, Ww = 8, 240, 320
    exp = {
        "general": {"name": "bench_cfg3", "clean_up_folder_if_exists": True,
                    "checkpoint_load": ""},
        "model": {"pretrained": False, "pretrained_backbone": False,
                  "num_classes": N_CLASSES, "backbone": args.backbone,
                  "amp": args.seg_amp},
        "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
        "trainer": {}, "data_module": {"batch_size": B},
        "scenes": ["scene0000_00"],
        "synthetic": {"n_views": 2 * B * max(1, world), "H": Hh, "W": Ww},
        "nerf": {"n_rays": 4096, "num_steps": 256, "upsample_steps": 256,
                 "precision": args.nerf_precision},
        "nerf_seed": 123, "seed": 123,
    }
    tmp = tempfile.mkdtemp()
    torch.manual_seed(123)
    # default: as scripts/train_joint.py sets it (a look-up in the shipped MIOpen
    # databases; the exhaustive search only without them or with --seg-find)
    from ucsa_neural_rendering_amd._miopen_db import default_cudnn_benchmark
    torch.backends.cudnn.benchmark = (True if args.seg_find else
                                      False if args.no_seg_find else default_cudnn_benchmark())
    model = JointTrainLightningNet(exp, {"results": tmp, "scannet": tmp})
    dm = JointTrainDataModule(exp)
    dm.setup()
    tr = Trainer(max_epochs=1, device=str(dev))
    tr._attach(model)
    if dist:
        udist.broadcast_parameters_(model)
        torch.manual_seed(123 + rank)
    model.train()
    model.joint_train = True
    batches = [tr._to_device(b) for b in dm.train_dataloader_joint()]
    n_b = len(batches)
    for i in range(max(1, args.warmup)):
        model.training_step(batches[i % n_b], 0)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.training_step(batches[i % n_b], 0)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, dev, backend)
    dt = elapsed / args.steps
    rays = world * B * (Hh * Ww + 4096)
    # roofline of the whole joint step: algorithmic flop / bytes of its three
    # parts over the step time.  rho is measured on 4096 rays of
    # the first frame of the first batch.
    roof = None
    if rank == 0:
        b0 = batches[0][1] if isinstance(batches[0], (tuple, list)) else batches[0]
        nb = model.nerf_model
        g = torch.Generator(device=dev).manual_seed(3)
        sel = torch.randperm(Hh * Ww, device=dev, generator=g)[:4096]
        S = 512
        rho = masked_fraction(nb, b0["rays_o"][0][sel], b0["rays_d"][0][sel],
                              b0["direction_norms"][0][sel], 256, 256, None,
                              torch.rand(4096, 256, device=dev, generator=g))
        per_sample = 6144 + rho * 19584
        f_render = B * Hh * Ww * S * per_sample
        f_train = B * 3.0 * 4096 * S * per_sample
        seg = model.seg_model
        seg.eval()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(args.seg_amp)):
            f_seg_fwd = conv_flops(seg, torch.rand(1, 3, Hh, Ww, device=dev))
        seg.train()
        f_seg = 3.0 * B * f_seg_fwd
        n_params = sum(p.numel() for p in nb.parameters())
        by_render = B * Hh * Ww * (204 + S * 1024.0)
        by_train = B * (2 * 4096 * S * 1024.0 + 28.0 * n_params)
        n_seg = sum(p.numel() for p in seg.parameters())
        by_seg = 28.0 * n_seg      # Adam only: activations are MIOpen's business
        flop = f_render + f_train + f_seg
        byts = by_render + by_train + by_seg
        roof = {
            "what": "one joint step per rank (ms_per_step)",
            "masked_fraction_rho": rho,
            "mfma": {"algorithmic_flop": flop,
                     "of_which": {"renders_8x320x240x512": f_render,
                                  "nerf_train_8x4096x512_fwd_bwd": f_train,
                                  "deeplab_fwd_bwd_8_images": f_seg},
                     "achieved_tflops": flop / dt / 1e12,
                     "frac_of_fp32_mfma_peak": flop / dt / 1e12 / F32_MFMA_PEAK_TF,
                     "frac_of_fp16_dense_peak": flop / dt / 1e12 / F16_MFMA_PEAK_TF},
            "hbm": {"algorithmic_bytes": byts,
                    "of_which": {"render_gathers_and_ray_io": by_render,
                                 "nerf_train_gather_scatter_adam": by_train,
                                 "deeplab_adam_28B_per_param": by_seg},
                    "achieved_gbs": byts / dt / 1e9,
                    "frac": byts / dt / 1e9 / HBM_PEAK_GBS},
        }
    result = {
        "metric": "rays/sec", "value": rays / dt, "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg3: joint step, 8 frames 320x240 per rank: 8 x (full "
                               "no-grad render 256+256 + 4096-ray NeRF train step) + "
                               f"DeepLabV3-{args.backbone} fwd/bwd/Adam on [8,3,240,320]",
                   "mode": "cfg3", "backbone": args.backbone,
                   "seg_precision": args.seg_amp or "fp32",
                   "nerf_render_nets": args.nerf_precision,
                   "nerf_rays_per_step_per_rank": B * (Hh * Ww + 4096),
                   "seg_images_per_s": world * B / dt,
                   "optimizer_nerf": type(model.optimizers()[1]).__name__},
        "losses": {k: v for k, v in model.logged.items()},
        "roofline_step": roof,
    }
    finish(dist, rank, result)

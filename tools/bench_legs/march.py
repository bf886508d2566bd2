"""Occupancy-grid marching leg on the benchmarked field."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import *  # noqa: F401,F403


def march_option(net, scene_ds, rays, n_views, out_live, dev, args):
    """Same field rendered by occupancy-grid marching (run_cuda, segmented
    schedule, far closure): rays/s, points per ray, quality of the last view
    against the analytic ground truth and against the live render."""
    from ucsa_neural_rendering_amd import ops
    from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
        SemanticNeRFNetwork
    from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
    m = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=True,
                            density_scale=1, num_semantic_classes=N_CLASSES,
                            seed=123).to(dev).eval()
    m.load_state_dict(net.state_dict(), strict=False)
    t0 = time.perf_counter()
    m.update_extra_state()
    torch.cuda.synchronize()
    grid_ms = (time.perf_counter() - t0) * 1e3
    res = {"density_grid_update_ms": grid_ms, "mean_density": m.mean_density,
           "dt_gamma": 1 / 128, "march_caps": [32, 96, 1024], "w_min": 1e-4}
    n = min(5, args.steps)
    for prec in ("fp32", "f16x2", "fp16"):
        m.precision = prec
        with torch.no_grad():
            for i in range(2):
                m.run_cuda(*rays[i], dt_gamma=1 / 128)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n):
                o = m.run_cuda(*rays[n_views - n + i], dt_gamma=1 / 128)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / n
        _, gt_rgb, gt_lab = scene_ds.room.cast(rays[n_views - 1][0][0],
                                               rays[n_views - 1][1][0])
        meter = SemanticsMeter(N_CLASSES)
        meter.update(o["semantics"][0].argmax(-1), gt_lab)
        res[prec] = {
            "rays_per_s": H * W / dt, "ms_per_view": dt * 1e3,
            "points_per_ray": m.last_march_points / (H * W),
            "rounds": m.last_march_rounds,
            "psnr_db": float(-10 * torch.log10(torch.mean((o["image"][0] - gt_rgb) ** 2))),
            "miou": meter.measure()[0],
            "max_abs_image_diff_vs_live": float((o["image"] - out_live["image"]).abs().max()),
        }
    # the intended use: a field trained THROUGH the marcher (same number of
    # Adam steps as the headline field), rendered by it without far closure
    # at least 400 steps: a fresh field needs ~150 before its air is empty
    # (density-grid decay), and the last 100 are timed separately
    steps = max(int(args.pretrain_steps), 400) if args.pretrain_steps > 0 else 0
    if steps > 0:
        from ucsa_neural_rendering_amd import losses as ul
        from ucsa_neural_rendering_amd.nerf.optim import HipAdam
        t = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=True,
                                density_scale=1, seed=123,
                                num_semantic_classes=N_CLASSES).to(dev).train()
        t.march_training = True
        opt = HipAdam(
            [{"name": "encoding", "params": list(t.encoder.parameters())},
             {"name": "net", "params": list(t.sigma_net.parameters()) +
              list(t.color_net.parameters()) +
              list(t.semantics_net.parameters()), "weight_decay": 1e-6}],
            lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        g = torch.Generator(device=dev).manual_seed(123)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_tail = None
        for it in range(steps):
            if it == steps - 100:
                torch.cuda.synchronize()
                t_tail = time.perf_counter()
            if t.refresh_due(it):
                t.update_extra_state()
            item = scene_ds[it % len(scene_ds)]
            inds = torch.randint(0, 240 * 320, (4096,), device=dev, generator=g)
            o = t.render(item["rays_o"][inds][None], item["rays_d"][inds][None],
                         item["direction_norms"][inds][None], perturb=True,
                         dt_gamma=1 / 256)
            lc, ls, ld = ul.nerf_losses(
                o["image"], o["semantics"], o["depth"],
                item["img"].reshape(3, -1).t()[inds][None],
                item["label"].reshape(-1)[inds][None],
                item["depth"].float().reshape(-1)[inds][None], 1.0)
            loss = ul.nerf_total_loss(lc, ls, ld)
            opt.zero_grad()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        dt_train = (time.perf_counter() - t0) / steps
        dt_tail = (time.perf_counter() - t_tail) / 100
        t.eval()
        t.update_extra_state()
        by_prec = {}
        for prec in ("f16x2", "fp16", "fp32"):      # (fp32 last: `o`, `dt` below are its)
            t.precision = prec
            with torch.no_grad():
                for i in range(2):
                    t.run_cuda(*rays[i], dt_gamma=1 / 256, far_closure=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n):
                    o = t.run_cuda(*rays[n_views - n + i], dt_gamma=1 / 256,
                                   far_closure=False)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / n
            by_prec[prec] = {"render_rays_per_s": H * W / dt, "render_ms_per_view": dt * 1e3,
                             "psnr_db": float(-10 * torch.log10(torch.mean((o["image"][0] - gt_rgb) ** 2)))}
        meter = SemanticsMeter(N_CLASSES)
        meter.update(o["semantics"][0].argmax(-1), gt_lab)
        res["trained_through_marcher"] = {
            "train_steps": steps, "train_ms_per_step": dt_train * 1e3,
            "train_rays_per_s": 4096 / dt_train,
            "train_ms_per_step_last_100": dt_tail * 1e3,
            "train_rays_per_s_last_100": 4096 / dt_tail, "dt_gamma": 1 / 256,
            "render_rays_per_s": H * W / dt, "render_ms_per_view": dt * 1e3,
            "render_by_arithmetic": by_prec,
            "points_per_ray": t.last_march_points / (H * W),
            "psnr_db": float(-10 * torch.log10(torch.mean((o["image"][0] - gt_rgb) ** 2))),
            "miou": meter.measure()[0],
            "note": "fresh field, same seed as the headline field, "
                    "max(pretrain_steps, 400) Adam steps of 4096 rays through "
                    "the marcher, rendered by it (no far closure); compare "
                    "train_ms_per_step_last_100 with train.ms_per_step and "
                    "psnr_db/miou with `quality`"}
    res["note"] = ("run_cuda on the field of the headline run: grid refresh, "
                   "segmented march (exact spans, device-side alive count), "
                   "hash encode + sigma MLP on the marched points, fused "
                   "weights/compaction/shading, far closure; compare "
                   "psnr_db/miou with `quality`")
    return res

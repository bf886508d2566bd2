"""`bench.py --mode cfg4`: the 512-view render job."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import *  # noqa: F401,F403


def cfg4_job(net, n_views, rank, world, dev, dist=None, backend="nccl", warmup=1,
             gather=False, precision="f16x2", keep=(), views_per_call=4):
    """BASELINE cfg4's render job: `n_views` novel 640x480 views round-robin
    over the ranks (this rank renders views rank, rank+world, ...), per view
    get_rays (a1) + staged render at 96+96 samples, parameters replicated, no
    data-path collective.  Returns (max-over-ranks seconds, views of this
    rank, {view index: rays + outputs} for the indices in `keep` that this
    rank rendered -- used by tests/test_gpu_configs.py for the oracle spot
    checks).  `views_per_call` views go through ONE render call, as the
    reference's forward_nerf_test renders its whole batch of frames at once
    (joint_train_lightning_net.py:225-257): the pipelined call then fills /
    drains once per group instead of once per view (bit-identical images)."""
    from ucsa_neural_rendering_amd import dist as udist, ops
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    mine = udist.shard_round_robin(n_views, rank, world)
    poses = _slerp_loop_poses(n_views, seed=999)[mine].to(dev)
    net.precision = precision
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    u = torch.rand(H * W, T_FINE, device=dev, generator=g)
    kept_views = {}

    V = max(1, int(views_per_call))
    uV = u.repeat(V, 1) if V > 1 else u          # the same uniforms for every view

    def views(i0, n, record=()):
        o, d, nrm = ops.get_rays(poses[i0:i0 + n], intr, H, W)   # a1 inside the job
        with torch.no_grad():
            out = net.render(o.reshape(1, n * H * W, 3), d.reshape(1, n * H * W, 3),
                             nrm.reshape(1, n * H * W), staged=True, perturb=False,
                             num_steps=T_COARSE, upsample_steps=T_FINE,
                             rng_u=uV[:n * H * W], image_width=W)
        out = {k: v.reshape(n, H * W, *v.shape[2:]) for k, v in out.items()
               if torch.is_tensor(v) and v.shape[:2] == (1, n * H * W)}
        for j in range(n):
            if mine[i0 + j] in record:
                kept_views[mine[i0 + j]] = dict(
                    o=o[j:j + 1], d=d[j:j + 1], nrm=nrm[j:j + 1], u=u,
                    **{k: v[j:j + 1] for k, v in out.items()})
        return out

    for i in range(0, min(warmup, len(mine)), V):
        views(i, min(V, len(mine) - i))
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = []
    for i in range(0, len(mine), V):
        n = min(V, len(mine) - i)
        out = views(i, n, record=keep)
        if gather:
            kept.extend((out["image"][j] * 255).to(torch.uint8) for j in range(n))
    if gather and dist:
        loc = torch.stack(kept) if kept else torch.empty(0, H * W, 3, dtype=torch.uint8, device=dev)
        if backend != "nccl":
            loc = loc.cpu()
        sizes = [len(udist.shard_round_robin(n_views, r, world)) for r in range(world)]
        udist.gather_rows(loc, sizes)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, dev, backend)
    return elapsed, mine, kept_views


def main_cfg4(args, net, scene_ds, dev, dist, world, rank, backend, prelog):
    """--mode cfg4: `--views` novel 640x480 views round-robin over the ranks
    (BASELINE cfg4: 512), parameters replicated, no data-path collective;
    `--gather` additionally collects the images on rank 0 inside the timed
    region (the only collective a render job can need)."""
    elapsed, mine, _ = cfg4_job(net, args.views, rank, world, dev, dist, backend,
                                warmup=args.warmup, gather=args.gather,
                                precision=args.nerf_precision)
    views_by_rank = [list(map(int, mine))]
    if dist:
        views_by_rank = [None] * world
        dist.all_gather_object(views_by_rank, list(map(int, mine)))
    result = {
        "metric": "rays/sec", "value": args.views * H * W / elapsed,
        "unit": "rays/s", "n_gpus": world, "steps": args.views,
        "warmup": args.warmup, "ms_per_step": elapsed / max(1, len(mine)) * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg4: {args.views} novel views x 640x480 x 192 "
                               "samples/ray, views round-robin over the ranks, "
                               "get_rays + render per group of 4 views (one call, as "
                               "forward_nerf_test renders its batch of frames)",
                   "mode": "cfg4", "views_per_rank": len(mine),
                   "views_by_rank": views_by_rank,
                   "gather_to_rank0": bool(args.gather), "pretrain": prelog,
                   "mlp_arithmetic": MLP_ARITHMETIC[args.nerf_precision],
                   "total_s": elapsed},
    }
    finish(dist, rank, result)

"""NeRF training legs (cfg3's NeRF half): single-GPU step time with its
roofline, the data-parallel step, `bench.py --mode train`."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import *  # noqa: F401,F403
from .common import _tick, nerf_optimizer


def nerf_train_roofline(n_rays, S, rho, n_params, ms, pmc=None):
    """fwd + bwd = 3 x the forward flop; bytes = the forward
    gathers (L x 8 corners x F x 4 B = 1024 B per sample, fp32 table), the
    same amount scattered into the gradient table by the backward, and Adam's
    28 B per parameter (read p, g, m, v; write p, m, v)."""
    flop = 3.0 * n_rays * S * (6144 + rho * 19584)
    gather = n_rays * S * 1024.0
    adam = 28.0 * n_params
    byts = 2 * gather + adam
    r = {
        "what": "NeRF training step (fwd + bwd + Adam)",
        "masked_fraction_rho": rho,
        "mfma": {"algorithmic_flop": flop, "achieved_tflops": flop / ms / 1e9,
                 "peak_fp32_mfma_tflops": F32_MFMA_PEAK_TF,
                 "frac_of_fp32_mfma_peak": flop / ms / 1e9 / F32_MFMA_PEAK_TF,
                 "frac_of_fp16_dense_peak": flop / ms / 1e9 / F16_MFMA_PEAK_TF},
        "hbm": {"algorithmic_bytes": byts,
                "of_which": {"forward_gathers": gather, "backward_scatter": gather,
                             "adam_28B_per_param": adam},
                "achieved_gbs": byts / ms / 1e6, "peak_gbs": HBM_PEAK_GBS,
                "frac": byts / ms / 1e6 / HBM_PEAK_GBS},
        "bound": "neither line is close: the step is a chain of ~25 launches "
                 "(gather, MFMA, scatter, Adam phases in turn), each bound by its "
                 "own resource (DESIGN 5)",
    }
    if pmc:
        r["traffic"] = pmc
    return r


def train_throughput(net, ds, device, steps=20, n_rays=4096, T=256, t=256,
                     train_precision="fp32"):
    """cfg3's NeRF half at the reference's native sizes: 4096 rays x (256+256)
    samples, forward + backward + Adam per step."""
    from ucsa_neural_rendering_amd import losses as ul, ops
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    import copy
    net = copy.deepcopy(net).train()
    net.train_precision = train_precision
    opt = HipAdam(
        [{"name": "encoding", "params": list(net.encoder.parameters())},
         {"name": "net", "params": list(net.sigma_net.parameters()) +
          list(net.color_net.parameters()) +
          list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
        lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device=device).manual_seed(7)
    item = ds[0]
    inds = torch.randint(0, 240 * 320, (n_rays,), device=device, generator=g)
    inds = ops.tile_order(inds, 320, H=240)  # as JointTrainLightningNet.get_rays_train does
    o, d, nrm = item["rays_o"][inds][None], item["rays_d"][inds][None], item["direction_norms"][inds][None]
    gt_rgb = item["img"].reshape(3, -1).t()[inds][None]
    labels = item["label"].reshape(-1)[inds][None]
    gt_depth = item["depth"].float().reshape(-1)[inds][None]
    rt = torch.rand(n_rays, T, device=device, generator=g)
    ru = torch.rand(n_rays, t, device=device, generator=g)

    def one():
        out = net.render(o, d, nrm, perturb=True, num_steps=T, upsample_steps=t,
                         rng_t=rt, rng_u=ru)
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"],
                                    gt_rgb, labels, gt_depth, 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        loss.backward()
        opt.step()

    import gc
    for _ in range(3):
        one()
    # three timed blocks of `steps` steps, the median reported: a collection
    # of the previous legs' deep-copied fields (hipFree synchronises) landing
    # inside one block once doubled a leg's figure
    gc.collect()
    blocks = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / steps)
    dt = sorted(blocks)[1]
    rho = masked_fraction(net, o, d, nrm, T, t, rt, ru)
    n_params = sum(p.numel() for p in net.parameters())
    pmc = None
    try:
        pj = json.load(open(os.path.join(ROOT, TRAIN_PMC_JSON)))
        pmc = pj.get(train_precision)
    except (OSError, ValueError):
        pass
    return {"workload": f"NeRF train step, {n_rays} rays x ({T}+{t}) samples, "
                        "fwd+bwd+Adam (reference native sizes; the 4096 random "
                        "pixels are handed over tile-ordered, ops.tile_order)",
            "ms_per_step": dt * 1e3, "rays_per_s": n_rays / dt,
            "ms_per_step_blocks": [b * 1e3 for b in blocks],
            "roofline": nerf_train_roofline(n_rays, T + t, rho, n_params, dt * 1e3, pmc)}


def dp_train_leg(net, ds, dev, dist, world, rank, backend, steps=20, warmup=3,
                 n_rays=4096, T=256, t=256, replicated=False, comm_dtype=None,
                 eval_view=False, fresh=False, train_precision="bf16x3"):
    """The data-parallel NeRF training step north_star describes (reference
    DDP site scripts/train_joint.py:137-142, step
    joint_train_lightning_net.py:497-513): every rank draws ITS OWN `n_rays`
    pixels of ITS OWN frame, forward + backward on the HIP path, then the
    NeRF-parameter gradients are averaged over RCCL -- reduce-scatter + Adam on
    a 1/N slice + all-gather (ShardedHipAdam) or one all-reduce + replicated
    Adam -- weak scaling: `value` = world x n_rays / step time."""
    import copy
    from ucsa_neural_rendering_amd import dist as udist, losses as ul, ops
    if fresh:   # train from the initialisation instead of the pre-trained field
        from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
            SemanticNeRFNetwork
        net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=False,
                                  density_scale=1, num_semantic_classes=N_CLASSES,
                                  seed=123).to(dev).train()
    else:
        net = copy.deepcopy(net).train()
    # the arithmetic JointTrainLightningNet trains with by default (`nerf: {train_precision:
    # bf16x3}`, lightning/joint_train_lightning_net.py) -- the bare network class defaults to
    # the f32-input-MFMA path, which this leg ran until round 6 (5.0 instead of 3.2 ms per step)
    net.train_precision = train_precision
    opt = nerf_optimizer(net, world, replicated, comm_dtype)
    g = torch.Generator(device=dev).manual_seed(7 + rank)      # rank-specific draws
    params = list(net.parameters())

    def one(it):
        item = ds[(it * world + rank) % len(ds)]                # rank-specific frame
        inds = torch.randint(0, 240 * 320, (n_rays,), device=dev, generator=g)
        inds = ops.tile_order(inds, 320, H=240)
        out = net.render(item["rays_o"][inds][None], item["rays_d"][inds][None],
                         item["direction_norms"][inds][None], perturb=True,
                         num_steps=T, upsample_steps=t,
                         rng_t=torch.rand(n_rays, T, device=dev, generator=g),
                         rng_u=torch.rand(n_rays, t, device=dev, generator=g))
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"],
                                    item["img"].reshape(3, -1).t()[inds][None],
                                    item["label"].reshape(-1)[inds][None],
                                    item["depth"].float().reshape(-1)[inds][None], 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        loss.backward()
        if not getattr(opt, "handles_collectives", False):
            udist.average_grads_(params)
        opt.step()
        return loss

    for it in range(warmup):
        one(it)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        loss = one(warmup + it)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    res = {}
    elapsed = max_over_ranks(elapsed, dist, dev, backend)
    if dist:
        # replicas must still be identical: compare a parameter checksum
        cs = torch.stack([p.detach().double().sum() for p in params])
        cs = cs.to(dev if backend == "nccl" else "cpu")
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["replicas_identical"] = bool(torch.equal(lo, hi))
        # the collectives alone, on gradient-sized buffers (k iterations)
        n_grid = net.encoder.params.numel()
        cdev = dev if backend == "nccl" else torch.device("cpu")
        buf = torch.zeros(n_grid, device=cdev)
        per = (n_grid // world) // 4 * 4
        shard = torch.zeros(per, device=cdev)

        def timed(fn, k=10):
            fn()
            if cdev.type == "cuda":
                torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(k):
                fn()
            if cdev.type == "cuda":
                torch.cuda.synchronize()
            return (time.perf_counter() - t1) / k * 1e3

        res["allreduce_ms"] = timed(lambda: dist.all_reduce(buf))
        res["reduce_scatter_allgather_ms"] = timed(lambda: (
            dist.reduce_scatter_tensor(shard, buf[:per * world]),
            dist.all_gather_into_tensor(buf[:per * world], shard)))
        # which slice of the hash grid each rank's Adam owns (ShardedHipAdam):
        # together they must tile the parameter (tests/test_gpu_bench_modes.py)
        mine = [list(x) for x in getattr(opt, "last_shards", [])]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        res["adam_shards"] = {"by_rank": everyone,
                              "params_total": int(sum(p.numel() for p in params)),
                              "grid_numel": int(n_grid)}
        res["collective_ranks"] = dist.get_world_size()
        res["collective_backend"] = dist.get_backend()
        res["grad_payload_bytes"] = n_grid * 4
    dt = elapsed / steps
    if eval_view:
        # quality of the trained replica on a held-out 320x240 view (how a
        # reduced-precision gradient payload shows up, if it does)
        from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
        net.eval()
        pose = _slerp_loop_poses(7, seed=4242)[3:4].to(dev)
        o, d, nrm = ops.get_rays(pose, (0.89 * 320, 0.89 * 320, 160.0, 120.0), 240, 320)
        with torch.no_grad():
            out = net.render(o, d, nrm, staged=True, num_steps=96, upsample_steps=96,
                             image_width=320)
        _, gt_rgb, gt_lab = ds.room.cast(o[0], d[0])
        res["eval_psnr_db"] = float(-10 * torch.log10(torch.mean((out["image"][0] - gt_rgb) ** 2)))
        res["eval_label_acc"] = float((out["semantics"][0].argmax(-1) == gt_lab).float().mean())
    res.update({
        "workload": f"data-parallel NeRF train step: {n_rays} rays x ({T}+{t}) "
                    "samples per rank (own frame, own pixels, tile-ordered), "
                    "fwd+bwd, gradient average over the ranks, Adam",
        "optimizer": type(opt).__name__ + ("" if comm_dtype is None else f"[{comm_dtype}]"),
        "train_precision": train_precision,
        "ms_per_step": dt * 1e3, "rays_per_s": world * n_rays / dt,
        "rays_per_step_total": world * n_rays, "final_loss": float(loss.detach()),
        "comm_bytes_per_step_per_rank": getattr(opt, "last_comm_bytes", None),
    })
    return res


def main_train(args, net, scene_ds, dev, dist, world, rank, backend, prelog,
               comm_dtype):
    """--mode train: `value` = rays/s trained by the data-parallel step."""
    tr = dp_train_leg(net, scene_ds, dev, dist, world, rank, backend,
                      steps=args.steps, warmup=args.warmup,
                      replicated=args.replicated_adam, comm_dtype=comm_dtype,
                      eval_view=True, fresh=args.fresh)
    result = {
        "metric": "rays/sec", "value": tr["rays_per_s"], "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": tr["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg3 NeRF half, data-parallel: " + tr["workload"],
                   "mode": "train", "pretrain": prelog,
                   "optimizer": tr["optimizer"]},
        "train_dp": tr,
    }
    finish(dist, rank, result)


def train_legs(result, net, scene_ds, dev, all_modes=False):
    """The single-GPU NeRF training step in the default arithmetic
    (`nerf: {train_precision: bf16x3}`: forward of the colour / semantics stage
    on the split pair with the bf16x3 nets, fp32-grade) and, with --detail,
    in the other three."""
    result["train"] = train_throughput(net, scene_ds, dev, train_precision="bf16x3")
    result["train"]["workload"] += (
        "; colour / semantics forward on the split pair with bf16x3 nets "
        "(fp32-grade), everything else fp32 (f32-input MFMA)")
    if not all_modes:
        return
    tm = train_throughput(net, scene_ds, dev, train_precision="fp32")
    tm["workload"] += "; forward on the fused f32-input-MFMA kernel (`nerf: {train_precision: fp32}`)"
    result["train_f32_mfma_forward"] = tm
    tf = train_throughput(net, scene_ds, dev, train_precision="fp16")
    tf["workload"] += ("; colour / semantics nets forward + backward on f16 MFMA "
                       "(`nerf: {train_precision: fp16}`), sigma net and grid fp32")
    result["train_f16_nets"] = tf
    tt = train_throughput(net, scene_ds, dev, train_precision="tcnn")
    tt["workload"] += ("; tiny-cuda-nn's numerics end to end (`nerf: {train_precision: "
                       "tcnn}`): fp16 table copy and features, all three nets on f16 "
                       "MFMA, half2 bin records; fp32 master parameters")
    result["train_tcnn_numerics"] = tt
    result["value_tcnn_numerics"] = {
        "render_rays_per_s": result.get("value_fp16_nets_fp16_table"),
        "train_rays_per_s": tt["rays_per_s"],
        "note": "the reference's own arithmetic (tiny-cuda-nn: fp16 table, fp16 nets, fp32 "
                "accumulate) next to the fp32-grade headline `value`"}

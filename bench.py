#!/usr/bin/env python3
"""Headline benchmark: Semantic-NeRF render throughput (rays/sec), BASELINE
cfg2 -- one 640x480 view per step, 192 samples/ray (96 coarse + 96 fine),
hash grid L=16, MLP width 64, 40 classes, fp32.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run, one rank per
GPU.  Rendering shards by view (rays are independent): every rank renders its
own 640x480 view per step with replicated parameters and NO data-path
collective ("scaling": "weak"); the only collectives are the timing barrier
and the max-over-ranks of the elapsed time.

One JSON line on rank 0; extra objects:
  roofline            dominant kernel (hash-grid gather; algorithmic bytes vs
                      the 8 TB/s HBM line, measured HBM traffic, the resource
                      that actually binds it)
  roofline_composite  colour/semantics MLPs: fp32-MFMA peak (the mode it runs
                      in) AND the FP16-dense peak SURVEY 8d prices the MLP
                      stage against
  roofline_step       whole view: algorithmic bytes and FLOP / ms_per_step
  cpu_baseline        the CPU oracle (oracle/, "port") timed on the host cores
                      on a bounded sample of the same workload
  train_dp            (N > 1) the data-parallel NeRF training step: every rank
                      its own 4096 rays, RCCL reduce-scatter / all-gather (or
                      all-reduce) of the gradients, Adam

Other modes (never the driver's default):
  --mode train   value = rays/s TRAINED (4096 rays x (256+256) per rank and
                 step, fwd + bwd + gradient collectives + Adam), weak scaling
  --mode cfg4    BASELINE cfg4: 512 novel views of 640x480 round-robin over the
                 ranks, no data-path collective, optional gather to rank 0
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 480, 640
T_COARSE, T_FINE = 96, 96
N_CLASSES = 40
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3  # fp32-input MFMA = fp32 vector peak
F16_MFMA_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA (SURVEY 8d's MLP roofline)
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
PMC_JSON = "profiles/r03_pmc_traffic.json"
TRAIN_PMC_JSON = "profiles/r03_train_pmc.json"
SEG_PMC_JSON = "profiles/r03_seg_pmc.json"
ENC_BINDING_JSON = "profiles/r03_encoder_binding.json"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=32768,
                    help="rays per pass of the CPU baseline (3 passes, median)")
    ap.add_argument("--pretrain-steps", type=int, default=200,
                    help="Adam steps on the synthetic scene before timing "
                         "(SURVEY 8d: 200)")
    ap.add_argument("--no-train-bench", action="store_true")
    ap.add_argument("--mode", choices=["render", "train", "cfg3", "cfg4"],
                    default="render")
    ap.add_argument("--backbone", default="resnet50",
                    help="cfg3: DeepLab backbone (BASELINE cfg3 says ResNet-50; "
                         "the reference's model is resnet101)")
    ap.add_argument("--seg-amp", default="", help="cfg3: '' (fp32, the reference) or bf16")
    ap.add_argument("--nerf-precision", default="bf16x3",
                    choices=["fp32", "bf16x3", "fp16"],
                    help="arithmetic of the three MLPs in the no-grad renders: "
                         "bf16x3 (default; fp32-grade on the bf16 MFMA pipe, "
                         "csrc/mfma_mlp_x3.h), fp32 (f32-input MFMA, an exact "
                         "fmaf chain), fp16 (tiny-cuda-nn's own numerics)")
    ap.add_argument("--seg-find", action="store_true",
                    help="DeepLab legs with torch.backends.cudnn.benchmark (MIOpen's "
                         "exhaustive solver search, what scripts/train_joint.py "
                         "uses): several minutes of search on a fresh box, "
                         "49 -> 44 ms per R-101 fp32 step")
    ap.add_argument("--no-seg-find", action="store_true",
                    help="cfg3: immediate-mode MIOpen solvers instead of the "
                         "module's default exhaustive search (profiling runs)")
    ap.add_argument("--replicated-adam", action="store_true",
                    help="train legs: all-reduce + full Adam on every rank "
                         "instead of the sharded optimizer")
    ap.add_argument("--grad-comm-dtype", choices=["fp32", "fp16", "bf16"],
                    default="fp32")
    ap.add_argument("--fresh", action="store_true",
                    help="--mode train: start from the initialisation (quality "
                         "comparisons of the gradient payload precision)")
    ap.add_argument("--views", type=int, default=512, help="cfg4: views in total")
    ap.add_argument("--gather", action="store_true",
                    help="cfg4: gather the images on rank 0 inside the timed region")
    return ap.parse_args()


_T0 = time.perf_counter()


def _tick(what):
    """Wall-clock log of the sections of a run (stderr; the JSON line stays
    alone on stdout)."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f} s] {what}", file=sys.stderr,
          flush=True)


def build_field(device, seed=123, train_steps=200, log=None, cuda_ray=False):
    """SURVEY 8d parameter state: tcnn-style init (grid U(-1e-4,1e-4), Xavier
    MLPs, seed 123), then `train_steps` Adam steps (lr 1e-2, the reference's
    NeRF optimizer) on the synthetic box-room scene so that sigma is
    non-trivial and the w > 1e-4 mask is selective.  Runs on the HIP training
    path; excluded from the timed region."""
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
    from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
        SemanticNeRFNetwork
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    from ucsa_neural_rendering_amd.ops import tile_order
    net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=cuda_ray,
                              density_scale=1, num_semantic_classes=N_CLASSES,
                              seed=seed).to(device).train()
    ds = SyntheticSceneDataset(0, n_views=16, H=240, W=320,
                               n_classes=N_CLASSES, device=device)
    opt = HipAdam(
        [{"name": "encoding", "params": list(net.encoder.parameters())},
         {"name": "net", "params": list(net.sigma_net.parameters()) +
          list(net.color_net.parameters()) +
          list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
        lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device=device).manual_seed(seed)
    t0 = time.perf_counter()
    for it in range(train_steps):
        item = ds[it % len(ds)]
        inds = tile_order(torch.randint(0, 240 * 320, (4096,), device=device, generator=g), 320, H=240)
        o, d, nrm = item["rays_o"][inds], item["rays_d"][inds], item["direction_norms"][inds]
        gt_rgb = item["img"].reshape(3, -1).t()[inds][None]
        labels = item["label"].reshape(-1)[inds][None]
        gt_depth = item["depth"].float().reshape(-1)[inds][None]
        out = net.render(o[None], d[None], nrm[None], perturb=True,
                         num_steps=T_COARSE, upsample_steps=T_FINE,
                         rng_t=torch.rand(4096, T_COARSE, device=device, generator=g),
                         rng_u=torch.rand(4096, T_FINE, device=device, generator=g))
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"],
                                    gt_rgb, labels, gt_depth, 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    if log is not None:
        log["pretrain_steps"] = train_steps
        log["pretrain_s"] = time.perf_counter() - t0
        log["pretrain_final_loss"] = float(loss.detach())
    return net.eval(), ds


def masked_fraction(net, o, d, nrm, T, t, rt, ru):
    """rho of a ray batch: fraction of the T + t samples per ray whose weight
    passes the reference's mask w > 1e-4 (renderer_semantics.py:249-250) --
    the samples the colour / semantics nets run on.  Staged ops, the
    composite's own aux weights."""
    from ucsa_neural_rendering_amd import ops
    with torch.no_grad():
        f = net._field()
        aabb = net._aabb_list(net.training)
        o, d, nrm = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous(), nrm.reshape(-1).contiguous()
        N = o.shape[0]
        near, far = ops.near_far_from_aabb(o, d, aabb)
        zc = ops.sample_coarse(near, far, T, rt)
        hc, sc = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb),
                                   f["packed_sigma"])
        zf = ops.resample(zc, sc.view(N, T), ru)
        hf, sf = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb),
                                   f["packed_sigma"])
        w = ops.composite_fwd(d, nrm, zc, sc.view(N, T), hc, zf, sf.view(N, t), hf,
                              f["packed_color"], f["packed_sem"], N_CLASSES, 1.0, want_aux=True)[4]
        return float((w > 1e-4).float().mean())


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


def conv_flops(model, x):
    """Forward FLOP of the convolutions (2 x MACs) and linear layers of
    `model` on input `x`, counted with forward hooks on the build's own
    modules ('FLOPs from a counter on the build's own model')."""
    total = [0]
    hooks = []

    def conv_hook(m, inp, out):
        kh, kw = m.kernel_size
        total[0] += 2 * out.numel() * (m.in_channels // m.groups) * kh * kw

    def lin_hook(m, inp, out):
        total[0] += 2 * out.numel() * m.in_features

    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, torch.nn.Linear):
            hooks.append(m.register_forward_hook(lin_hook))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return total[0]


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


def _nerf_optimizer(net, world, replicated=False, comm_dtype=None):
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam, ShardedHipAdam
    groups = [{"name": "encoding", "params": list(net.encoder.parameters())},
              {"name": "net", "params": list(net.sigma_net.parameters()) +
               list(net.color_net.parameters()) +
               list(net.semantics_net.parameters()), "weight_decay": 1e-6}]
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    if world > 1 and not replicated:
        return ShardedHipAdam(groups, comm_dtype=comm_dtype, **kw)
    return HipAdam(groups, **kw)


def dp_train_leg(net, ds, dev, dist, world, rank, backend, steps=20, warmup=3,
                 n_rays=4096, T=256, t=256, replicated=False, comm_dtype=None,
                 eval_view=False, fresh=False):
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
    opt = _nerf_optimizer(net, world, replicated, comm_dtype)
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
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
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
        "ms_per_step": dt * 1e3, "rays_per_s": world * n_rays / dt,
        "rays_per_step_total": world * n_rays, "final_loss": float(loss.detach()),
        "comm_bytes_per_step_per_rank": getattr(opt, "last_comm_bytes", None),
    })
    return res


def seg_throughput(device, steps=5, B=8, find=False):
    """cfg3's segmentation half: DeepLabV3-ResNet-101 forward + backward +
    Adam on [8,3,240,320] uniform-random images / labels, with the
    reference's CE-on-softmax loss through ucsa_seg_tail.

    * ``fp32``: the module's default path -- fp32 like the reference (no
      autocast around seg), channels-last, every BatchNorm (+ add) (+ ReLU) one
      fused HIP op (csrc/batchnorm.hip), 1x1 convolutions as one GEMM over the
      batch; 3x3 / 7x7 convolutions are MIOpen.
    * ``fp32_nchw_unfused``: the same modules on NCHW inputs, i.e.
      F.batch_norm + add + relu kernels and MIOpen's per-image 1x1 GEMMs (what
      rounds 1-2 measured as "fp32").
    * ``bf16_channels_last``: bf16 autocast, fused BatchNorm in bf16.
    * ``*_graph``: forward and backward replayed as HIP graphs
      (torch.cuda.make_graphed_callables); the optimizer step stays eager."""
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.network import DeepLabV3
    out = {}
    # MIOpen exhaustive find, as scripts/train_joint.py sets it: minutes of
    # search on a fresh box, so the default bench run measures immediate mode
    before = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(find)
    out["miopen_find"] = bool(find)
    for mode in ("fp32", "fp32_nchw_unfused", "bf16_channels_last", "fp32_graph",
                 "bf16_graph"):
        torch.manual_seed(0)
        amp = mode.startswith("bf16")
        nchw = mode == "fp32_nchw_unfused"
        m = DeepLabV3({"pretrained": False, "pretrained_backbone": False,
                       "num_classes": N_CLASSES}).to(device).train()
        x = torch.rand(B, 3, 240, 320, device=device)
        if not nchw:
            m = m.to(memory_format=torch.channels_last)
            x = x.contiguous(memory_format=torch.channels_last)
        y = torch.randint(-1, N_CLASSES, (B, 240, 320), device=device)
        opt = torch.optim.Adam(m.parameters(), lr=1e-5, fused=not nchw)

        class _Net(torch.nn.Module):   # parameters visible to make_graphed_callables
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, inp):
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    return self.inner(inp)["out"]

        net = _Net(m)
        fwd = net
        try:
            if mode.endswith("_graph"):
                fwd = torch.cuda.make_graphed_callables(net, (x.clone(),))
        except Exception as e:  # report, do not hide
            out[mode] = {"failed": repr(e)[:300]}
            del m, opt
            torch.cuda.empty_cache()
            continue

        def one():
            logits = fwd(x)
            loss = ul.seg_loss(logits.float().contiguous(), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss

        for _ in range(2):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if "fwd_flop_per_image" not in out:
            m.eval()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out["fwd_flop_per_image"] = conv_flops(m, x[:1])
            m.train()
        flop = 3.0 * B * out["fwd_flop_per_image"]
        peak = F16_MFMA_PEAK_TF if amp else F32_MFMA_PEAK_TF
        out[mode] = {"ms_per_step": dt * 1e3, "images_per_s": B / dt,
                     "loss": float(loss),
                     "roofline": {"bound": "mfma", "algorithmic_flop": flop,
                                  "achieved": flop / dt / 1e12, "peak": peak,
                                  "unit": "TFLOP/s", "frac": flop / dt / 1e12 / peak,
                                  "note": "3 x the forward convolution flop of the "
                                          "mirror (hook counter, conv_flops) x 8 images "
                                          "/ step time; peak = " +
                                          ("bf16 dense MFMA" if amp else
                                           "fp32-input MFMA (= fp32 vector) rate")}}
        # HBM traffic / MFMA-busy per step from the committed PMC passes
        # (tools/seg_pmc.sh; rocprofv3 cannot run inside this process)
        key = {"fp32": "fp32_cl", "bf16_channels_last": "bf16_cl"}.get(mode)
        if key:
            try:
                pj = json.load(open(os.path.join(ROOT, SEG_PMC_JSON))).get(key)
            except (OSError, ValueError):
                pj = None
            if pj:
                r = out[mode]["roofline"]
                r["traffic"] = pj["hbm_bytes_per_step"]
                r["traffic_source"] = SEG_PMC_JSON
                r["hbm_utilisation"] = pj["hbm_bytes_per_step"] / dt / 1e9 / HBM_PEAK_GBS
                r["mfma_pipe_busy_frac"] = pj["mfma_busy_frac"]
                r["valu_issue_frac"] = pj["valu_issue_frac"]
        del m, opt, fwd
        torch.cuda.empty_cache()
    out["workload"] = ("DeepLabV3-ResNet-101 train step, batch 8 x 3x240x320, "
                       "CE-on-softmax loss, Adam")
    torch.backends.cudnn.benchmark = before
    return out


def effective_cores() -> int:
    """Cores this process may actually use: min(cpu_count, affinity, cgroup
    quota).  (The GPU box shows 256 CPUs but a 16-CPU cgroup quota; 256 OpenMP
    threads on 16 CPUs do not finish.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(net, pose, intr, n_rays, threads):
    """Time the CPU oracle on `n_rays` rays of the same view, same T/t."""
    from oracle import field as ofield
    from oracle import rays as orays
    from oracle import renderer as oren
    torch.set_num_threads(threads)
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=N_CLASSES,
                             seed=None)
    fld.grid_params = net.encoder.params.detach().cpu()
    fld.sigma_params = net.sigma_net.params.detach().cpu()
    fld.color_params = net.color_net.params.detach().cpu()
    fld.sem_params = net.semantics_net.params.detach().cpu()
    o, d, nrm = orays.pixel_rays(pose[None].cpu(), intr, H, W)
    g = torch.Generator().manual_seed(5)
    sel = torch.randperm(H * W, generator=g)[:n_rays]
    o, d, nrm = o[:, sel], d[:, sel], nrm[:, sel]
    u = torch.rand(n_rays, T_FINE, generator=g)
    aabb = torch.tensor([-4.0, -4, -4, 4, 4, 4])
    times = []
    with torch.no_grad():
        for it in range(3):   # >= 3 repeats, median (~8 s per pass)
            t0 = time.perf_counter()
            ref = oren.run(fld, o, d, nrm, aabb, num_steps=T_COARSE,
                           upsample_steps=T_FINE, u=u)
            times.append(time.perf_counter() - t0)
        best = sorted(times)[1]
        # the same rays / uniforms through the HIP path: parity + matched PSNR
        dev = net.encoder.params.device
        got = net.render(o.to(dev), d.to(dev), nrm.to(dev), num_steps=T_COARSE,
                         upsample_steps=T_FINE, rng_u=u.to(dev))
    parity = {k: float((got[k].cpu() - ref[k]).abs().max())
              for k in ("image", "depth", "semantics")}
    return n_rays / best, times, parity, ref, got, (o, d)


MLP_ARITHMETIC = {
    "bf16x3": "fp32-grade on the bf16 MFMA pipe: every fp32 weight and layer "
              "input split exactly into three bf16 terms, six partial products "
              "per product (v_mfma_f32_16x16x32_bf16), fp32 accumulation "
              "(csrc/mfma_mlp_x3.h); within 1-2 ulp of the f32-input MFMA mode "
              "(f32_mfma_option.max_abs_image_diff_vs_value_mode)",
    "fp32": "f32-input MFMA (v_mfma_f32_16x16x4_f32): bit for bit a k-ordered "
            "fmaf chain",
    "fp16": "tiny-cuda-nn's own numerics: fp16 weights / layer inputs, fp32 "
            "accumulation (v_mfma_f32_16x16x32_f16)",
}


def composite_roofline(mode, mlp_tf, sig_tf, launch_ms):
    """MFMA roofline object of the colour + semantics stage (algorithmic flop
    of the masked samples / launch time)."""
    if mode == "fp32":
        return {
            "kernel": "k_composite (colour+semantics MLPs, fp32 MFMA)",
            "bound": "mfma", "achieved": mlp_tf, "peak": F32_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": mlp_tf / F32_MFMA_PEAK_TF,
            "frac_of_fp16_dense_peak": mlp_tf / F16_MFMA_PEAK_TF,
            "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
            "note": "peak = fp32-input MFMA (the instruction this mode issues, "
                    "1/16 of the 16-bit rate; it runs at the vector FMA rate "
                    "and, measured, does not overlap with VALU work at all: "
                    "kernel time = MFMA busy + VALU issue); "
                    "frac_of_fp16_dense_peak is the same achieved rate against "
                    "SURVEY 8d's 2.5 PF line"}
    if mode == "bf16x3":
        return {
            "kernel": "k_weights_compact + k_shade16<bf16x3> (colour + "
                      "semantics MLPs, six bf16 MFMA passes per fp32 product)",
            "bound": "mfma", "achieved": mlp_tf, "peak": F16_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": mlp_tf / F16_MFMA_PEAK_TF,
            "issued_mfma_tflops": mlp_tf * 6 * 22528 / 19584,
            "issued_frac": mlp_tf * 6 * 22528 / 19584 / F16_MFMA_PEAK_TF,
            "frac_of_fp32_mfma_peak": mlp_tf / F32_MFMA_PEAK_TF,
            "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
            "note": "achieved = ALGORITHMIC fp32 flop of the masked samples / "
                    "launch time against the 2.5 PF 16-bit dense line "
                    "(SURVEY 8d); issued_* counts the six bf16 passes and the "
                    "padding (144 MFMAs per 16 samples).  Measured (PMC, "
                    "profiles/r03_shade16_pmc.txt): kernel time = MFMA-busy "
                    "cycles + VALU issue cycles, the two do not overlap on a "
                    "SIMD shared by several waves"}
    return {
        "kernel": "k_weights_compact + k_shade16<f16> (colour+semantics "
                  "MLPs on 16x16x32 f16 MFMA, fp32 accumulate)",
        "bound": "mfma", "achieved": mlp_tf, "peak": F16_MFMA_PEAK_TF,
        "unit": "TFLOP/s", "frac": mlp_tf / F16_MFMA_PEAK_TF,
        "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
        "note": "the nets are 24 MFMAs per 16 samples here: the kernel is "
                "bound by VALU issue (softmax, conversions, ordered per-ray "
                "sums), not by the matrix pipe"}


def mlp_error_vs_fp64(net, dev, M=16384):
    """Max error of the colour / class-probability outputs of the three
    shading arithmetics against an fp64 evaluation of the same nets (torch,
    double, on the device): M random samples, one per ray, through
    ucsa_composite_infer with T = 1 and a huge density (weight 1), so the
    composite returns the nets' outputs themselves.  Shows in the bench line
    that bf16x3 is as close to fp64 as the exact f32-input MFMA chain."""
    from ucsa_neural_rendering_amd import ops
    C = N_CLASSES
    g = torch.Generator(device=dev).manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1)
    h = torch.randn(M, 16, device=dev, generator=g)
    cp, sp = net.color_net.params.detach(), net.semantics_net.params.detach()
    # fp64 reference: SH-4 of the direction mapped as the reference does
    x, y, z = [(((d[:, i].double() + 1) / 2) * 2 - 1) for i in range(3)]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    sh = torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y,
        0.48860251190291987 * z, -0.48860251190291987 * x, 1.0925484305920792 * xy,
        -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], dim=-1)
    geo = h[:, 1:].double()
    one = torch.ones(M, 1, dtype=torch.float64, device=dev)
    cpd, spd = cp.double(), sp.double()
    w1, w2, w3 = cpd[:2048].view(64, 32), cpd[2048:6144].view(64, 64), cpd[6144:7168].view(16, 64)
    xin = torch.cat([sh, geo, one], -1)
    rgb64 = torch.sigmoid(torch.relu(torch.relu(xin @ w1.t()) @ w2.t()) @ w3.t())[:, :3]
    out_pad = (C + 15) // 16 * 16
    s1, s2 = spd[:1024].view(64, 16), spd[1024:1024 + out_pad * 64].view(out_pad, 64)
    p64 = torch.softmax((torch.relu(torch.cat([geo, one], -1) @ s1.t()) @ s2.t())[:, :C], -1)
    zc = torch.ones(M, 1, device=dev)
    sg = torch.full((M, 1), 50.0, device=dev)
    nrm = torch.ones(M, device=dev)
    args = (d, nrm, zc, sg, h, None, None, None)
    res = {}
    for name, pc, ps, kw in (
            ("f32_mfma", ops.mlp_pack(1, cp), ops.mlp_pack(2, sp, C), {}),
            ("bf16x3", ops.mlp_pack_x3(1, cp), ops.mlp_pack_x3(2, sp, C), {"x3": True}),
            ("fp16", ops.mlp_pack_f16(1, cp), ops.mlp_pack_f16(2, sp, C), {"half": True})):
        img, _, sem = ops.composite_infer(*args, pc, ps, C, **kw)
        res[name] = {"rgb": float((img.double() - rgb64).abs().max()),
                     "class_probability": float((sem.double() - p64).abs().max())}
    res["note"] = ("max |kernel - fp64| over %d random samples on the benchmarked "
                   "field's colour / semantics nets" % M)
    return res


def stage_times(net, o, d, nrm, u, iters=5, image_width=0, half=False,
                mode=None):
    """Per-kernel durations of one chunk, measured with events on the stream
    the kernels run on (torch's current stream), launched the way
    ucsa_render_fwd[_f16|_x3] launches them.  mode: "fp32" (f32-input MFMA,
    fused composite), "fp16" / "bf16x3" (sigma MLP and the split composite
    pair on the 16-bit MFMA pipe)."""
    from ucsa_neural_rendering_amd import ops
    mode = mode or ("fp16" if half else "fp32")
    half = mode == "fp16"
    x3 = mode == "bf16x3"
    f = net._field_f16() if half else (net._field_x3() if x3 else net._field())
    sigma_mlp = (ops.sigma_mlp_fwd_f16 if half else
                 ops.sigma_mlp_fwd_x3 if x3 else ops.sigma_mlp_fwd)
    aabb = net._aabb_list(False)
    N = o.shape[0]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    names = ["near_far+coarse", "encode_c", "sigma_c", "resample", "encode_f",
             "sigma_f", "composite"]
    acc = {k: 0.0 for k in names}
    rho = 0.0
    for it in range(iters + 1):
        marks = [ev() for _ in range(len(names) + 1)]
        marks[0].record()
        near, far = ops.near_far_from_aabb(o, d, aabb)
        zc = ops.sample_coarse(near, far, T_COARSE)
        marks[1].record()
        feat = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb,
                                        image_width=image_width, half_features=half)
        marks[2].record()
        hc, sc = sigma_mlp(feat, f["packed_sigma"])
        marks[3].record()
        zf = ops.resample(zc, sc.view(N, T_COARSE), u)
        marks[4].record()
        feat = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb,
                                        image_width=image_width, half_features=half)
        marks[5].record()
        hf, sf = sigma_mlp(feat, f["packed_sigma"])
        marks[6].record()
        if it == 0:   # the weights, for the masked fraction rho
            f32 = net._field()
            w = ops.composite_fwd(d, nrm, zc, sc.view(N, T_COARSE), hc, zf,
                                  sf.view(N, T_FINE), hf, f32["packed_color"],
                                  f32["packed_sem"], N_CLASSES, 1.0, want_aux=True)[4]
        elif half or x3:
            ops.composite_infer(d, nrm, zc, sc.view(N, T_COARSE), hc, zf,
                                sf.view(N, T_FINE), hf, f["packed_color"],
                                f["packed_sem"], N_CLASSES, 1.0, half=half, x3=x3)
        else:   # what ucsa_render_fwd launches for fp32: the fused kernel
            ops.composite_fwd(d, nrm, zc, sc.view(N, T_COARSE), hc, zf,
                              sf.view(N, T_FINE), hf, f["packed_color"],
                              f["packed_sem"], N_CLASSES, 1.0)
        marks[7].record()
        torch.cuda.synchronize()
        if it == 0:
            rho = float((w > 1e-4).float().mean())
            continue
        for i, k in enumerate(names):
            acc[k] += marks[i].elapsed_time(marks[i + 1]) / iters
    return acc, rho


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
    for prec in ("fp32", "fp16"):
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
        meter = SemanticsMeter(N_CLASSES)
        meter.update(o["semantics"][0].argmax(-1), gt_lab)
        res["trained_through_marcher"] = {
            "train_steps": steps, "train_ms_per_step": dt_train * 1e3,
            "train_rays_per_s": 4096 / dt_train,
            "train_ms_per_step_last_100": dt_tail * 1e3,
            "train_rays_per_s_last_100": 4096 / dt_tail, "dt_gamma": 1 / 256,
            "render_rays_per_s": H * W / dt, "render_ms_per_view": dt * 1e3,
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


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start
    the N ranks as a CHILD `python -m torch.distributed.run` and relay its
    stdout / exit code.  This process never touches the GPU (no torch.cuda
    call that initialises HIP happens before this point;
    ``torch.cuda.device_count()`` does not) -- a process that has initialised
    the GPU must not exec or be replaced, so the ranks are children and the
    parent only waits.  The reference's DDP site: scripts/train_joint.py:137-142
    (Lightning spawns the ranks there)."""
    import socket
    import subprocess
    backend = os.environ.get("UCSA_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        raise SystemExit(
            f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s). One rank "
            "per GPU over RCCL needs that many; UCSA_BENCH_BACKEND=gloo runs all "
            "ranks on cuda:0 (code-path check on a 1-GPU box, not a measurement)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["UCSA_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    _tick("self-launch: " + " ".join(cmd[1:]))
    # stdout through a FILE, not a pipe: helper processes the ranks leave
    # behind for a while would keep an inherited pipe open past torchrun's exit
    import tempfile
    with tempfile.TemporaryFile("w+") as fo:
        proc = subprocess.Popen(cmd, env=env, stdout=fo, stdin=subprocess.DEVNULL)
        rc = proc.wait()
        fo.seek(0)
        sys.stdout.write(fo.read())   # rank 0's JSON line (stderr passed through)
        sys.stdout.flush()
    return rc


def dist_record(dist, world, rank, backend, dev, args):
    """What actually ran: world size as torch.distributed sees it, backend,
    the device of every rank.  Fails loudly when the process group has a
    different number of ranks than --gpus."""
    launcher = os.environ.get("UCSA_BENCH_LAUNCHER", "torchrun" if world > 1 else "none")
    if dist is None:
        return {"world_size": 1, "backend": None, "launcher": launcher,
                "devices": [f"{dev} ({torch.cuda.get_device_name(dev)})"]}
    ws = dist.get_world_size()
    if ws != args.gpus:
        raise SystemExit(f"process group has {ws} ranks, --gpus {args.gpus}")
    mine = (rank, str(dev), torch.cuda.get_device_name(dev))
    devs = [None] * ws
    dist.all_gather_object(devs, mine)
    if backend == "nccl" and len({d for _, d, _ in devs}) != ws:
        raise SystemExit(f"RCCL ranks share a device: {devs}")
    devs = [f"rank {r}: {d} ({n})" for r, d, n in devs]
    return {"world_size": ws, "backend": dist.get_backend() + (" (RCCL)" if backend == "nccl" else ""),
            "launcher": launcher, "devices": devs}


_DIST_RECORD = None


def main():
    if os.environ.get("UCSA_BENCH_WATCHDOG"):   # debugging aid: stacks of a stuck rank
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["UCSA_BENCH_WATCHDOG"]), exit=True)
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # test hook: UCSA_BENCH_BACKEND=gloo runs all ranks on cuda:0 (the 1-GPU
    # dev box) to exercise the N>1 code path; the driver's real runs use RCCL
    backend = os.environ.get("UCSA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    global _DIST_RECORD
    _DIST_RECORD = dist_record(dist, world, rank, backend, dev, args)

    from ucsa_neural_rendering_amd import ops
    if args.mode == "cfg3":
        return main_cfg3(args, dev, dist, world, rank, backend)
    prelog = {}
    _tick("imports, device")
    net, scene_ds = build_field(dev, train_steps=args.pretrain_steps, log=prelog)
    _tick("field pre-trained")
    if dist:
        # the pre-training is not bit-reproducible (float atomics in the grid
        # backward): all ranks render / train rank 0's field
        from ucsa_neural_rendering_amd import dist as udist_
        udist_.broadcast_parameters_(net)
    net.hip_ray_chunk = 65536
    comm_dtype = {"fp32": None, "fp16": torch.float16,
                  "bf16": torch.bfloat16}[args.grad_comm_dtype]
    if args.mode == "train":
        return main_train(args, net, scene_ds, dev, dist, world, rank, backend,
                          prelog, comm_dtype)
    if args.mode == "cfg4":
        return main_cfg4(args, net, scene_ds, dev, dist, world, rank, backend, prelog)
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    n_views = args.steps + args.warmup
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
    poses = _slerp_loop_poses(n_views * world, seed=999)[rank::world].to(dev)
    # inputs resident in HBM before the timed region: rays of every view this
    # rank renders, and the uniforms for the inverse-CDF resampling
    rays = [ops.get_rays(poses[i:i + 1], intr, H, W) for i in range(n_views)]
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    u = torch.rand(H * W, T_FINE, device=dev, generator=g)

    def step(i):
        o, d, nrm = rays[i]
        with torch.no_grad():
            return net.render(o, d, nrm, staged=True, perturb=False,
                              num_steps=T_COARSE, upsample_steps=T_FINE,
                              rng_u=u, image_width=W)

    net.precision = args.nerf_precision
    for i in range(args.warmup):
        step(i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    rays_total = world * args.steps * H * W
    value = rays_total / elapsed

    result = None
    if rank == 0:
        assert torch.isfinite(out["image"]).all()
        # the chunk render() really launches: whole 8-row bands of the image
        chunk = net.hip_ray_chunk - net.hip_ray_chunk % (8 * W)
        o, d, nrm = rays[0]
        st, rho = stage_times(net, o[0, :chunk].contiguous(),
                              d[0, :chunk].contiguous(),
                              nrm[0, :chunk, 0].contiguous(), u[:chunk],
                              image_width=W, mode=args.nerf_precision)
        # --- roofline of the dominant kernel: hash-grid encode -------------
        # algorithmic bytes per sample: L * 8 corners * F * 4 B
        samples = chunk * T_COARSE
        enc_bytes = samples * 16 * 8 * 2 * 4
        enc_ms = 0.5 * (st["encode_c"] + st["encode_f"])
        enc_gbs = enc_bytes / (enc_ms * 1e-3) / 1e9
        # --- MLP roofline: composite kernel (masked colour + semantics) ----
        masked = rho * chunk * (T_COARSE + T_FINE)
        mlp_flop = masked * (12544 + 7040)
        mlp_tf = mlp_flop / (st["composite"] * 1e-3) / 1e12
        sig_tf = samples * 6144 / (0.5 * (st["sigma_c"] + st["sigma_f"]) * 1e-3) / 1e12
        ms_step = elapsed / args.steps * 1e3
        S = T_COARSE + T_FINE
        # whole view, SURVEY 8d / BASELINE.md 2.4 definitions
        step_bytes_fp32 = H * W * (204 + S * 1024)        # this build: fp32 table
        step_bytes_fp16 = H * W * (204 + S * 512)         # the definition's fp16 table
        step_flop = H * W * S * (6144 + rho * 19584)
        result = {
            "metric": "rays/sec",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg2: Semantic-NeRF render, one 640x480 view per "
                            "step per GPU, 192 samples/ray (96 coarse + 96 "
                            "fine), hash grid L=16 F=2 T=2^19, MLP width 64, "
                            "40 classes",
                "rays_per_step_per_gpu": H * W,
                "ray_chunk": chunk,
                "parameter_state": "tcnn-style init (seed 123) + %d Adam steps "
                                   "on the synthetic box-room scene (SURVEY 8d)"
                                   % prelog.get("pretrain_steps", 0),
                "pretrain": prelog,
                "masked_fraction_rho": rho,
                "sharding": "views round-robin over ranks, no data-path collective",
                "timed_region": "net.render() of one view per step (rows a2-a10); "
                                "ray generation (a1, ucsa_get_rays, ~10 us per view) "
                                "runs BEFORE the timed region: its outputs are the "
                                "HBM-resident inputs of the step (bench.py --mode cfg4 "
                                "times get_rays + render per view)",
                "mlp_arithmetic": MLP_ARITHMETIC[args.nerf_precision],
            },
            "roofline_encode": {
                "kernel": "k_hashgrid_encode_tiled",
                "bound": "hbm",
                "achieved": enc_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": enc_gbs / HBM_PEAK_GBS,
                "traffic": None,
                "launch_ms": enc_ms,
                "algorithmic_bytes_per_launch": enc_bytes,
                "note": "achieved = ALGORITHMIC gather bytes (1024 B/sample) / "
                        "launch time: a nominal figure against the HBM line, "
                        "NOT HBM utilisation (hbm_utilisation below is: measured "
                        "traffic / time / peak).  The table slice a launch phase "
                        "works on is L2/MALL resident; what binds the kernel is "
                        "the L2->L1 path of divergent gathers (binding_resource)",
            },
            "roofline_composite": composite_roofline(args.nerf_precision, mlp_tf,
                                                     sig_tf, st["composite"]),
            "roofline_step": {
                "what": "one 640x480 view end to end (ms_per_step)",
                "hbm": {"algorithmic_bytes_fp32_table": step_bytes_fp32,
                        "achieved_gbs_fp32_table": step_bytes_fp32 / ms_step / 1e6,
                        "frac_fp32_table": step_bytes_fp32 / ms_step / 1e6 / HBM_PEAK_GBS,
                        "algorithmic_bytes_fp16_table_definition": step_bytes_fp16,
                        "frac_fp16_table_definition":
                            step_bytes_fp16 / ms_step / 1e6 / HBM_PEAK_GBS,
                        "peak_gbs": HBM_PEAK_GBS},
                "mfma": {"algorithmic_flop": step_flop,
                         "achieved_tflops": step_flop / ms_step / 1e9,
                         "frac_of_fp16_dense_peak":
                             step_flop / ms_step / 1e9 / F16_MFMA_PEAK_TF,
                         "frac_of_fp32_mfma_peak":
                             step_flop / ms_step / 1e9 / F32_MFMA_PEAK_TF,
                         "peak_fp16_dense_tflops": F16_MFMA_PEAK_TF},
            },
            "stage_ms_per_chunk": st,
        }
        # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot
        # run inside this process): only when they were collected on the same
        # parameter state (pretrain steps) as this run, else null
        try:
            pmc = json.load(open(os.path.join(ROOT, PMC_JSON)))
            if int(pmc.get("pretrain_steps", -1)) == int(args.pretrain_steps):
                cmp_kernel = {"fp32": "k_composite", "bf16x3": "k_shade16_x3",
                              "fp16": "k_shade16_f16"}[args.nerf_precision]
                for key, kn in (("roofline_composite", cmp_kernel),
                                ("roofline_encode", "k_hashgrid_encode_tiled")):
                    if kn not in pmc or "fetch_bytes" not in pmc[kn]:
                        if kn in pmc and "mfma_busy_frac" in pmc[kn]:
                            result[key]["mfma_pipe_busy_frac"] = pmc[kn]["mfma_busy_frac"]
                            result[key]["valu_issue_frac"] = pmc[kn].get("valu_issue_frac")
                            result[key]["pmc_source"] = PMC_JSON
                        continue
                    tr = pmc[kn]["fetch_bytes"] + pmc[kn]["write_bytes"]
                    if kn.startswith("k_shade16") and "k_weights_compact" in pmc:
                        # the stage is two launches: weights / compaction, nets
                        tr += (pmc["k_weights_compact"]["fetch_bytes"] +
                               pmc["k_weights_compact"]["write_bytes"])
                    r = result[key]
                    r["traffic"] = tr
                    r["traffic_source"] = PMC_JSON
                    r["hbm_utilisation"] = tr / (r["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                    if "tcc_hit_rate" in pmc[kn]:
                        r["tcc_hit_rate"] = pmc[kn]["tcc_hit_rate"]
                    if key == "roofline_composite" and "mfma_busy_frac" in pmc[kn]:
                        r["mfma_pipe_busy_frac"] = pmc[kn]["mfma_busy_frac"]
                    if "valu_issue_frac" in pmc[kn]:
                        r["valu_issue_frac"] = pmc[kn]["valu_issue_frac"]
                e = result["roofline_encode"]
                # what binds the encoder: the per-CU L1's (TCP) line look-up
                # rate, from the PMC passes of the kernel alone
                eb = json.load(open(os.path.join(ROOT, ENC_BINDING_JSON)))
                acc = 0.5 * (eb["coarse"]["tcp_accesses_per_clock_per_cu"] +
                             eb["fine"]["tcp_accesses_per_clock_per_cu"])
                e["binding_resource"] = {
                    "resource": "TCP (per-CU vector L1) line look-ups of divergent gathers, "
                                "1 per clock and CU",
                    "achieved": acc, "peak": 1.0, "unit": "line accesses / clock / CU",
                    "frac": acc,
                    "fine_pass": {k: eb["fine"][k] for k in (
                        "tcp_accesses_per_clock_per_cu", "l1_hit_rate", "l2_latency_cycles",
                        "misses_in_flight_per_tcp", "tcp_pending_stall_frac",
                        "valu_issue_frac", "l2_request_frac_of_34500")},
                    "coarse_pass": {k: eb["coarse"][k] for k in (
                        "tcp_accesses_per_clock_per_cu", "l1_hit_rate", "l2_latency_cycles",
                        "misses_in_flight_per_tcp", "tcp_pending_stall_frac",
                        "valu_issue_frac", "l2_request_frac_of_34500")},
                    "source": ENC_BINDING_JSON + " (tools/encode_pmc.sh)"}
                if "k_shade16_f16" in pmc and "mfma_busy_frac" in pmc["k_shade16_f16"]:
                    result["pmc_k_shade16_f16"] = {
                        "mfma_pipe_busy_frac": pmc["k_shade16_f16"]["mfma_busy_frac"],
                        "valu_issue_frac": pmc["k_shade16_f16"].get("valu_issue_frac"),
                        "valu_wave_instructions_per_launch":
                            pmc["k_shade16_f16"].get("valu_wave_instructions"),
                        "source": PMC_JSON}
        except OSError as e:
            raise SystemExit(f"bench.py: {PMC_JSON} (the committed PMC passes the "
                             f"roofline's `traffic` comes from) is missing: {e}")
        except (KeyError, ValueError) as e:
            raise SystemExit(f"bench.py: {PMC_JSON} is malformed: {e!r}")
        # "roofline" = the kernel with the largest share of the step
        enc_share = st["encode_c"] + st["encode_f"]
        dom = "roofline_encode" if enc_share >= st["composite"] else "roofline_composite"
        result["roofline"] = dict(result[dom])
        # quality of the timed renders: last view vs the analytic ground truth
        t_hit, gt_rgb, gt_lab = scene_ds.room.cast(rays[n_views - 1][0][0], rays[n_views - 1][1][0])
        mse = torch.mean((out["image"][0] - gt_rgb) ** 2)
        _, pred_lab = ops.semantic_postproc(out["semantics"][0], want_normalised=False)
        from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
        meter = SemanticsMeter(N_CLASSES)
        meter.update(pred_lab, gt_lab)
        result["quality"] = {"psnr_db": float(-10 * torch.log10(mse)),
                             "miou": meter.measure()[0],
                             "note": "novel 640x480 view vs analytic GT after "
                                     "the pre-training above"}
        # the same workload in the other arithmetic modes of the three MLPs
        # (never the headline `value`, which is --nerf-precision's mode)
        ref_img = step(args.warmup + args.steps - 1)["image"]
        n_alt = min(5, args.steps)
        for alt in ("fp32", "bf16x3", "fp16"):
            if alt == args.nerf_precision:
                continue
            net.precision = alt
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n_alt):
                step(args.warmup + i)
            torch.cuda.synchronize()
            dta = (time.perf_counter() - t1) / n_alt
            diff = (step(args.warmup + args.steps - 1)["image"] - ref_img).abs().max()
            sta, _ = stage_times(net, o[0, :chunk].contiguous(),
                                 d[0, :chunk].contiguous(),
                                 nrm[0, :chunk, 0].contiguous(), u[:chunk],
                                 image_width=W, mode=alt)
            cmp_tf = mlp_flop / (sta["composite"] * 1e-3) / 1e12
            sga_tf = samples * 6144 / (0.5 * (sta["sigma_c"] + sta["sigma_f"]) * 1e-3) / 1e12
            key = {"fp32": "f32_mfma_option", "bf16x3": "bf16x3_option",
                   "fp16": "f16_mlp_option"}[alt]
            vkey = {"fp32": "value_f32_mfma_nets", "bf16x3": "value_bf16x3_nets",
                    "fp16": "value_fp16_nets"}[alt]
            result[vkey] = world * H * W / dta if world == 1 else None
            result[key] = {
                "rays_per_s": H * W / dta, "ms_per_view": dta * 1e3,
                "max_abs_image_diff_vs_value_mode": float(diff),
                "stage_ms_per_chunk": sta,
                "roofline_composite": composite_roofline(alt, cmp_tf, sga_tf,
                                                         sta["composite"]),
                "roofline_step_mfma_frac_of_fp16_dense_peak":
                    step_flop / (dta * 1e3) / 1e9 / F16_MFMA_PEAK_TF,
                "mlp_arithmetic": MLP_ARITHMETIC[alt],
                "select": "`nerf: {precision: %s}` / --nerf-precision %s" % (alt, alt)}
        # fp16 nets AND the hash grid read from an fp16 copy of the table: what
        # tiny-cuda-nn stores and computes with (`nerf: {precision: fp16,
        # fp16_table: true}`)
        net.precision, net.fp16_table = "fp16", True
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_alt):
            step(args.warmup + i)
        torch.cuda.synchronize()
        dth = (time.perf_counter() - t1) / n_alt
        diff_h = (step(args.warmup + args.steps - 1)["image"] - ref_img).abs().max()
        net.fp16_table = False
        result["value_fp16_nets_fp16_table"] = world * H * W / dth if world == 1 else None
        result["fp16_table_option"] = {
            "rays_per_s": H * W / dth, "ms_per_view": dth * 1e3,
            "max_abs_image_diff_vs_value_mode": float(diff_h),
            "note": "fp16 nets + half2 hash table (26 MB instead of 52 MB; fp32 "
                    "master copy with the optimizer): 4-byte entries let one "
                    "16-byte access serve an aligned group of four x-neighbours, "
                    "5 instead of 6 accesses per sample and hashed level; parity: "
                    "features bit-identical to the fp32 kernels on the rounded "
                    "table, render vs the oracle with rounded table + fp16 nets "
                    "(tests/test_gpu_parity.py::test_fp16_table_*)",
            "select": "`nerf: {precision: fp16, fp16_table: true}`"}
        if "f16_mlp_option" in result:
            result["f16_mlp_option"]["note"] = (
                "parity of this mode: tests/test_gpu_configs.py against the "
                "oracle with tiny-cuda-nn's roundings emulated (3e-3)")
        net.precision = args.nerf_precision
        # The side measurements below (marcher, training, DeepLab, CPU
        # baseline) are single-GPU figures: at N > 1 the other ranks would only
        # wait for rank 0, so they run at N = 1 only.
        extras = world == 1
        # occupancy-grid marching on the same parameters:
        # never the headline `value` (cfg2 is defined at 192 samples/ray)
        result["mlp_error_vs_fp64"] = mlp_error_vs_fp64(net, dev)
        _tick("render modes measured")
        if extras:
            try:
                result["march_option"] = march_option(net, scene_ds, rays,
                                                      n_views, out, dev, args)
            except Exception as e:  # the headline line must survive, loudly
                import traceback
                traceback.print_exc(file=sys.stderr)
                result["march_option"] = {"error": repr(e), "failed": True}
        _tick("marcher option done")
        if extras and not args.no_train_bench:
            # default training arithmetic (`nerf: {train_precision: bf16x3}`):
            # forward of the colour / semantics stage on the split pair with
            # the bf16x3 nets (fp32-grade), backward on the f32-input MFMA
            result["train"] = train_throughput(net, scene_ds, dev, train_precision="bf16x3")
            result["train"]["workload"] += (
                "; colour / semantics forward on the split pair with bf16x3 nets "
                "(fp32-grade), everything else fp32 (f32-input MFMA)")
            tm = train_throughput(net, scene_ds, dev, train_precision="fp32")
            tm["workload"] += "; forward on the fused f32-input-MFMA kernel (`nerf: {train_precision: fp32}`)"
            result["train_f32_mfma_forward"] = tm
            tf = train_throughput(net, scene_ds, dev, train_precision="fp16")
            tf["workload"] += ("; colour / semantics nets forward + backward on f16 "
                               "MFMA (`nerf: {train_precision: fp16}`), sigma net and "
                               "grid fp32")
            result["train_f16_nets"] = tf
            tt = train_throughput(net, scene_ds, dev, train_precision="tcnn")
            tt["workload"] += ("; tiny-cuda-nn's numerics end to end (`nerf: {train_precision: "
                               "tcnn}`): fp16 table copy and features, all three nets on f16 "
                               "MFMA, half2 bin records; fp32 master parameters")
            result["train_tcnn_numerics"] = tt
            result["value_tcnn_numerics"] = {
                "render_rays_per_s": result.get("value_fp16_nets_fp16_table"),
                "train_rays_per_s": tt["rays_per_s"],
                "note": "the reference's own arithmetic (tiny-cuda-nn: fp16 table, fp16 "
                        "nets, fp32 accumulate) next to the fp32-grade headline `value`; "
                        "parity of both modes against the oracle with those roundings "
                        "emulated: tests/test_gpu_configs.py (render), "
                        "tests/test_gpu_backward.py::test_tcnn_numerics_* (training)"}
            _tick("training legs done")
            result["seg"] = seg_throughput(dev, find=args.seg_find)
            _tick("DeepLab leg done")
        if extras and not args.no_cpu_baseline:
            threads = effective_cores()
            v, dt, parity, ref, got, (co, cd) = cpu_baseline(
                net, poses[0], intr, args.cpu_rays, threads)
            _, gt_c, _ = scene_ds.room.cast(co[0].to(dev), cd[0].to(dev))
            psnr = lambda img: float(-10 * torch.log10(torch.mean((img - gt_c.to(img.device)) ** 2)))
            result["cpu_vs_gpu"] = {
                "max_abs_diff": parity,
                "psnr_db_cpu": psnr(ref["image"][0]),
                "psnr_db_gpu": psnr(got["image"][0]),
            }
            result["cpu_baseline"] = {
                "value": v,
                "unit": "rays/s",
                "cores": threads,
                "kind": "port",
                "sample": f"{args.cpu_rays} random rays of view 0, same "
                          f"T={T_COARSE}/t={T_FINE}; median of 3 passes ("
                          + ", ".join(f"{x:.1f}" for x in dt) + " s)",
                "pass_seconds": dt,
            }
            result["speedup_vs_cpu"] = value / v
            _tick("CPU baseline done")
    if world > 1:
        # the data-parallel training step with its gradient collectives: all
        # ranks take part; rank 0 reports it next to the render line
        tr = dp_train_leg(net, scene_ds, dev, dist, world, rank, backend,
                          steps=min(args.steps, 20), replicated=args.replicated_adam,
                          comm_dtype=comm_dtype)
        if rank == 0:
            result["train_dp"] = tr
    if rank == 0:
        result["distributed"] = _DIST_RECORD
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def _finish(dist, rank, result):
    if rank == 0:
        result["distributed"] = _DIST_RECORD
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


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
    _finish(dist, rank, result)


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
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
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
    _finish(dist, rank, result)


def cfg4_job(net, n_views, rank, world, dev, dist=None, backend="nccl", warmup=1,
             gather=False, precision="bf16x3", keep=()):
    """BASELINE cfg4's render job: `n_views` novel 640x480 views round-robin
    over the ranks (this rank renders views rank, rank+world, ...), per view
    get_rays (a1) + staged render at 96+96 samples, parameters replicated, no
    data-path collective.  Returns (max-over-ranks seconds, views of this
    rank, {view index: rays + outputs} for the indices in `keep` that this
    rank rendered -- used by tests/test_gpu_configs.py for the oracle spot
    checks).  Reference: forward_nerf_test, joint_train_lightning_net.py:225-257."""
    from ucsa_neural_rendering_amd import dist as udist, ops
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    mine = udist.shard_round_robin(n_views, rank, world)
    poses = _slerp_loop_poses(n_views, seed=999)[mine].to(dev)
    net.precision = precision
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    u = torch.rand(H * W, T_FINE, device=dev, generator=g)
    kept_views = {}

    def view(i, record=False):
        o, d, nrm = ops.get_rays(poses[i:i + 1], intr, H, W)   # a1 inside the job
        with torch.no_grad():
            out = net.render(o, d, nrm, staged=True, perturb=False,
                             num_steps=T_COARSE, upsample_steps=T_FINE,
                             rng_u=u, image_width=W)
        if record:
            kept_views[mine[i]] = dict(o=o, d=d, nrm=nrm, u=u, **out)
        return out

    for i in range(min(warmup, len(mine))):
        view(i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = []
    for i in range(len(mine)):
        out = view(i, record=mine[i] in keep)
        if gather:
            kept.append((out["image"][0] * 255).to(torch.uint8))
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
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, mine, kept_views


def main_cfg4(args, net, scene_ds, dev, dist, world, rank, backend, prelog):
    """--mode cfg4: `--views` novel 640x480 views round-robin over the ranks
    (BASELINE cfg4: 512), parameters replicated, no data-path collective;
    `--gather` additionally collects the images on rank 0 inside the timed
    region (the only collective a render job can need)."""
    elapsed, mine, _ = cfg4_job(net, args.views, rank, world, dev, dist, backend,
                                warmup=args.warmup, gather=args.gather,
                                precision=args.nerf_precision)
    result = {
        "metric": "rays/sec", "value": args.views * H * W / elapsed,
        "unit": "rays/s", "n_gpus": world, "steps": args.views,
        "warmup": args.warmup, "ms_per_step": elapsed / max(1, len(mine)) * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg4: {args.views} novel views x 640x480 x 192 "
                               "samples/ray, views round-robin over the ranks, "
                               "get_rays + render per view",
                   "mode": "cfg4", "views_per_rank": len(mine),
                   "gather_to_rank0": bool(args.gather), "pretrain": prelog,
                   "mlp_arithmetic": MLP_ARITHMETIC[args.nerf_precision],
                   "total_s": elapsed},
    }
    _finish(dist, rank, result)


if __name__ == "__main__":
    main()

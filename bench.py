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
  roofline      hash-grid encode kernel (dominant, HBM/gather bound)
  roofline_mlp  composite kernel's colour/semantics MLPs (fp32 MFMA bound)
  cpu_baseline  the CPU oracle (oracle/, "port") timed on the host cores on a
                bounded sample of the same workload
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=49152)
    return ap.parse_args()


def look_at_pose(eye, target, up=(0.0, 0.0, 1.0)):
    """cam2world with +z forward (the reference's ray convention: dir=(x,y,1))."""
    eye = torch.tensor(eye, dtype=torch.float32)
    f = torch.tensor(target, dtype=torch.float32) - eye
    f = f / f.norm()
    upv = torch.tensor(up, dtype=torch.float32)
    r = torch.linalg.cross(f, upv)
    r = r / r.norm()
    d = torch.linalg.cross(f, r)
    m = torch.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, d, f, eye
    return m


def synthetic_poses(n, seed=123):
    """Cameras on a loop inside the bound-4 box (|eye| < 2.5), looking at
    points near the centre -- every ray hits the box."""
    g = torch.Generator().manual_seed(seed)
    poses = []
    for k in range(n):
        a = 2 * torch.pi * (k / max(n, 1)) + 0.1
        eye = (2.0 * torch.cos(torch.tensor(a)).item(),
               2.0 * torch.sin(torch.tensor(a)).item(),
               0.3 * torch.randn(1, generator=g).item())
        tgt = (0.3 * torch.randn(1, generator=g).item(),
               0.3 * torch.randn(1, generator=g).item(), 0.0)
        poses.append(look_at_pose(eye, tgt))
    return torch.stack(poses)


def build_field(device, seed=123):
    """Seeded, untrained field with a lively grid (see DESIGN.md 'bench
    parameter state'): tcnn-style MLP init, grid ~ U(-3, 3)."""
    from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
        SemanticNeRFNetwork
    net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=False,
                              density_scale=1, num_semantic_classes=N_CLASSES,
                              seed=seed)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        net.encoder.params.copy_(
            (torch.rand(net.encoder.params.numel(), generator=g) * 2 - 1) * 3.0)
    return net.to(device).eval()


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
    best = None
    with torch.no_grad():
        for it in range(2):
            t0 = time.perf_counter()
            oren.run(fld, o, d, nrm, aabb, num_steps=T_COARSE,
                     upsample_steps=T_FINE, u=u)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    return n_rays / best, best


def stage_times(net, o, d, nrm, u, iters=5):
    """Per-kernel durations of one chunk, measured with events on the stream
    the kernels run on (torch's current stream)."""
    from ucsa_neural_rendering_amd import ops
    f = net._field()
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
        feat = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb)
        marks[2].record()
        hc, sc = ops.sigma_mlp_fwd(feat, f["packed_sigma"])
        marks[3].record()
        zf = ops.resample(zc, sc.view(N, T_COARSE), u)
        marks[4].record()
        feat = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb)
        marks[5].record()
        hf, sf = ops.sigma_mlp_fwd(feat, f["packed_sigma"])
        marks[6].record()
        img, dep, sem, src, w = ops.composite_fwd(
            d, nrm, zc, sc.view(N, T_COARSE), hc, zf, sf.view(N, T_FINE), hf,
            f["packed_color"], f["packed_sem"], N_CLASSES, 1.0, want_aux=True)
        marks[7].record()
        torch.cuda.synchronize()
        if it == 0:
            rho = float((w > 1e-4).float().mean())
            continue
        for i, k in enumerate(names):
            acc[k] += marks[i].elapsed_time(marks[i + 1]) / iters
    return acc, rho


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ucsa_neural_rendering_amd import ops
    net = build_field(dev)
    net.hip_ray_chunk = 32768
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    n_views = args.steps + args.warmup
    poses = synthetic_poses(n_views * world)[rank::world].to(dev)
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
                              rng_u=u)

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
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    rays_total = world * args.steps * H * W
    value = rays_total / elapsed

    result = None
    if rank == 0:
        assert torch.isfinite(out["image"]).all()
        chunk = net.hip_ray_chunk
        o, d, nrm = rays[0]
        st, rho = stage_times(net, o[0, :chunk].contiguous(),
                              d[0, :chunk].contiguous(),
                              nrm[0, :chunk, 0].contiguous(), u[:chunk])
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
        result = {
            "metric": "rays/sec",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
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
                "parameter_state": "seeded untrained field, grid U(-3,3)",
                "masked_fraction_rho": rho,
                "sharding": "views round-robin over ranks, no data-path collective",
            },
            "roofline_encode": {
                "kernel": "k_hashgrid_encode",
                "bound": "hbm",
                "achieved": enc_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": enc_gbs / HBM_PEAK_GBS,
                "traffic": None,
                "launch_ms": enc_ms,
                "algorithmic_bytes_per_launch": enc_bytes,
            },
            "roofline_composite": {
                "kernel": "k_composite (colour+semantics MLPs, fp32 MFMA)",
                "bound": "mfma",
                "achieved": mlp_tf,
                "peak": F32_MFMA_PEAK_TF,
                "unit": "TFLOP/s",
                "frac": mlp_tf / F32_MFMA_PEAK_TF,
                "launch_ms": st["composite"],
                "traffic": None,
                "sigma_mlp_tflops": sig_tf,
            },
            "stage_ms_per_chunk": st,
        }
        # "roofline" = the kernel with the largest share of the step
        enc_share = st["encode_c"] + st["encode_f"]
        dom = "roofline_encode" if enc_share >= st["composite"] else "roofline_composite"
        result["roofline"] = dict(result[dom])
        if not args.no_cpu_baseline:
            threads = effective_cores()
            v, dt = cpu_baseline(net, poses[0], intr, args.cpu_rays, threads)
            result["cpu_baseline"] = {
                "value": v,
                "unit": "rays/s",
                "cores": threads,
                "kind": "port",
                "sample": f"{args.cpu_rays} random rays of view 0, same "
                          f"T={T_COARSE}/t={T_FINE}, best of 2 ({dt:.1f} s each)",
            }
            result["speedup_vs_cpu"] = value / v
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""A/B of the experimental grid-backward switches (written at the end of round 5,
verified but NOT timed there): the merged grid backward of a training step alone
(HIP events, 4096 rays x (256+256) samples of the pre-trained bench field), then the
whole training step, with

    default            k_grid_bwd_bin<REC_P64> + k_grid_bwd_accum<REC_P64> + k_hashgrid_bwd<true>
    UCSA_BWD_XPAIR=1   one 16-byte record per x-pair of corners
    UCSA_BWD_XPAIR=2   ... and the run sums as DPP scans
    UCSA_BWD_DPP=1     the shipped kernels with DPP run plans / run sums
    UCSA_BWD_DPP=1 UCSA_BWD_XPAIR=2   both (coarse kernel DPP + x-pair DPP)

in ONE process on one box (the library reads the switches at every call), rounds
interleaved so that clock drift hits all of them alike.   python tools/bwd_switches_ab.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tools.bench_legs.train import train_throughput
from ucsa_neural_rendering_amd import ops

CONFIGS = [("default", {}), ("xpair=1", {"UCSA_BWD_XPAIR": "1"}), ("xpair=2", {"UCSA_BWD_XPAIR": "2"}),
           ("dpp=1", {"UCSA_BWD_DPP": "1"}), ("dpp=1 xpair=2", {"UCSA_BWD_DPP": "1", "UCSA_BWD_XPAIR": "2"})]


def use(env):
    for k in ("UCSA_BWD_XPAIR", "UCSA_BWD_DPP"):
        os.environ[k] = env.get(k, "0")


def main():
    dev = torch.device("cuda", 0)
    net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", "200")))
    f = net._field()
    aabb = net._aabb_list(True)
    N, Tc, Tf = 4096, 256, 256
    item = ds[0]
    g = torch.Generator(device=dev).manual_seed(3)
    inds = ops.tile_order(torch.randint(0, 240 * 320, (N,), device=dev, generator=g), 320, H=240)
    o, d = item["rays_o"][inds].contiguous(), item["rays_d"][inds].contiguous()
    near, far = ops.near_far_from_aabb(o, d, aabb)
    z = ops.sample_coarse(near, far, Tc, torch.rand(N, Tc, device=dev, generator=g))
    h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb), f["packed_sigma"])
    z_f = ops.resample(z, sig.view(N, Tc), torch.rand(N, Tf, device=dev, generator=g), 1.0)
    src = torch.sort(torch.cat([z, z_f], 1), dim=1, stable=True)[1].to(torch.int32).contiguous()
    d_c = torch.randn(16, N * Tc, 2, device=dev, generator=g) * 1e-3
    d_f = torch.randn(16, N * Tf, 2, device=dev, generator=g) * 1e-3
    grad = torch.zeros_like(net.encoder.params).view(-1, 2)
    times = {name: [] for name, _ in CONFIGS}
    for rnd in range(4):
        for name, env in CONFIGS:
            use(env)
            for _ in range(3):
                ops.hashgrid_bwd_rays_merged(f["grid"], o, d, z, z_f, src, aabb, d_c, d_f, grad, packed=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.hashgrid_bwd_rays_merged(f["grid"], o, d, z, z_f, src, aabb, d_c, d_f, grad, packed=True)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 10)
    print("merged grid backward alone (ms per call, 4 interleaved rounds):")
    for name, _ in CONFIGS:
        print(f"  {name:16s} " + " ".join(f"{t:.3f}" for t in times[name]) + f"   median {sorted(times[name])[len(times[name]) // 2]:.3f}")
    print("training step, 4096 x (256+256), bf16x3 (ms per step, 2 interleaved rounds):")
    step = {name: [] for name, _ in CONFIGS}
    for rnd in range(2):
        for name, env in CONFIGS:
            use(env)
            step[name].append(train_throughput(net, ds, dev, steps=20, train_precision="bf16x3")["ms_per_step"])
    for name, _ in CONFIGS:
        print(f"  {name:16s} " + " ".join(f"{t:.3f}" for t in step[name]))
    use({})


if __name__ == "__main__":
    main()

"""Marcher render time on a field trained THROUGH the marcher (the march_option leg of bench.py --detail,
without the rest of the bench): ms per 640x480 view and points per ray for each arithmetic.
STEPS=... training steps (default 400), TAG=... labels the lines."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tools.bench_legs.common import N_CLASSES
from ucsa_neural_rendering_amd import ops, losses as ul
from ucsa_neural_rendering_amd.nerf.optim import HipAdam
from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import SemanticNeRFNetwork
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses

dev = torch.device("cuda:0")
_, ds = bench.build_field(dev, train_steps=0)
W, H = 640, 480
poses = _slerp_loop_poses(8, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i + 1], (0.89 * W, 0.89 * W, W / 2, H / 2), H, W) for i in range(8)]
t = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=True, density_scale=1, seed=123,
                        num_semantic_classes=N_CLASSES).to(dev).train()
t.march_training = True
opt = HipAdam([{"name": "encoding", "params": list(t.encoder.parameters())},
               {"name": "net", "params": list(t.sigma_net.parameters()) + list(t.color_net.parameters()) +
                list(t.semantics_net.parameters()), "weight_decay": 1e-6}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
g = torch.Generator(device=dev).manual_seed(123)
for it in range(int(os.environ.get("STEPS", "400"))):
    if t.refresh_due(it):
        t.update_extra_state()
    item = ds[it % len(ds)]
    inds = torch.randint(0, 240 * 320, (4096,), device=dev, generator=g)
    o = t.render(item["rays_o"][inds][None], item["rays_d"][inds][None], item["direction_norms"][inds][None],
                 perturb=True, dt_gamma=1 / 256)
    lc, ls, ld = ul.nerf_losses(o["image"], o["semantics"], o["depth"], item["img"].reshape(3, -1).t()[inds][None],
                                item["label"].reshape(-1)[inds][None], item["depth"].float().reshape(-1)[inds][None], 1.0)
    loss = ul.nerf_total_loss(lc, ls, ld)
    opt.zero_grad(); loss.backward(); opt.step()
t.eval(); t.update_extra_state()
_, gt_rgb, _ = ds.room.cast(rays[7][0][0], rays[7][1][0])
for prec in os.environ.get("PRECS", "fp32,f16x2,fp16").split(","):
    t.precision = prec
    with torch.no_grad():
        for i in range(3):
            t.run_cuda(*rays[i], dt_gamma=1 / 256, far_closure=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(16):
            o = t.run_cuda(*rays[i % 8], dt_gamma=1 / 256, far_closure=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
        o = t.run_cuda(*rays[7], dt_gamma=1 / 256, far_closure=False)
    psnr = float(-10 * torch.log10(torch.mean((o["image"][0] - gt_rgb) ** 2)))
    print(f"{os.environ.get('TAG', '')} marcher {prec}: {dt * 1e3:.3f} ms/view -> {H * W / dt / 1e6:.2f} M rays/s; "
          f"{t.last_march_points / (H * W):.1f} points/ray, {t.last_march_rounds} rounds; PSNR {psnr:.2f} dB", flush=True)

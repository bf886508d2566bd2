"""cfg2 view time (f16x2, the bench's path) for the current environment; TAG=... labels the line.
W, H, T (samples per pass), FRAMES (views per call) change the workload: W=320 H=240 T=256 FRAMES=8 is the joint
step's render."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200, deterministic=True)
W, H = int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
T = int(os.environ.get("T", 96))
F = int(os.environ.get("FRAMES", 1))
poses = _slerp_loop_poses(8, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i + 1], (0.89 * W, 0.89 * W, W / 2, H / 2), H, W) for i in range(8)]
if F > 1:   # F views per call, as forward_nerf_test renders its batch of frames
    rays = [tuple(torch.cat([rays[(i + j) % 8][c] for j in range(F)], 1) for c in range(3)) for i in range(8)]
u = torch.rand(F * H * W, T, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
net.precision = "f16x2"
net.hip_ray_chunk = 65536
outs = []
for rep in range(3):
    with torch.no_grad():
        for i in range(3):
            net.render(*rays[i], num_steps=T, upsample_steps=T, rng_u=u, image_width=W)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(16):
            out = net.render(*rays[i % 8], num_steps=T, upsample_steps=T, rng_u=u, image_width=W)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
    outs.append(dt)
chk = float(out["image"].double().sum()), float(out["semantics"].double().sum()), float(out["depth"].double().sum())
print(f"{os.environ.get('TAG', '')}: " + " ".join(f"{d*1e3:.3f}" for d in outs) + f" ms/view  -> {F*H*W/min(outs)/1e6:.2f} M rays/s   checksum {chk}", flush=True)

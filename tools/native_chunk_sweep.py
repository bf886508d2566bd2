"""320x240 x (256+256) no-grad render (cfg3's native-size views) against the
ray chunk per enqueue (net.hip_ray_chunk), module-default arithmetic."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=200)
H, W, T, t = 240, 320, 256, 256
poses = _slerp_loop_poses(6, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i+1], (0.89*W, 0.89*W, W/2, H/2), H, W) for i in range(6)]
u = torch.rand(H*W, t, device=dev)
net.precision = "bf16x3"
for chunk in (65536, 38400, 25600, 76800, 65536):
    net.hip_ray_chunk = chunk
    with torch.no_grad():
        for i in range(2):
            net.render(*rays[i], staged=True, num_steps=T, upsample_steps=t, rng_u=u, image_width=W)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(12):
            net.render(*rays[i % 6], staged=True, num_steps=T, upsample_steps=t, rng_u=u, image_width=W)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
    print(f"chunk {chunk}: {dt*1e3:.2f} ms per 320x240x512 view, {H*W*(T+t)/dt/1e9:.2f} G samples/s", flush=True)

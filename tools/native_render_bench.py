"""Render throughput at the reference's native sizes: 320x240, 256+256 samples
(IMAGE_WIDTH=0: ray-ordered gather; default: image-ordered, as the module calls it)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=100)
H, W, T, t = 240, 320, 256, 256
IW = int(os.environ.get('IMAGE_WIDTH', str(W)))
if os.environ.get('CHUNK'):
    net.hip_ray_chunk = int(os.environ['CHUNK'])
poses = _slerp_loop_poses(6, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i+1], (0.89*W, 0.89*W, W/2, H/2), H, W) for i in range(6)]
u = torch.rand(H*W, t, device=dev)
for prec in ("fp32", "fp16"):
    net.precision = prec
    with torch.no_grad():
        for i in range(2):
            out = net.render(*rays[i], staged=True, num_steps=T, upsample_steps=t, rng_u=u, image_width=IW)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(6):
            out = net.render(*rays[i], staged=True, num_steps=T, upsample_steps=t, rng_u=u, image_width=IW)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"{prec}: {dt*1e3:.2f} ms per 320x240x512 view, {H*W/dt/1e6:.2f} M rays/s, {H*W*(T+t)/dt/1e9:.2f} G samples/s")
net.precision = "fp32"

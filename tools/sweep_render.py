"""Sweep ray-chunk size and stream count for the full-view render."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=100)
H, W = 480, 640
poses = _slerp_loop_poses(6, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i+1], (0.89*W, 0.89*W, W/2, H/2), H, W) for i in range(6)]
u = torch.rand(H*W, 96, device=dev)
ref = None
for chunk in (16384, 32768, 65536, 131072, 307200):
    for ns in (1, 2, 3):
        net.hip_ray_chunk, net.hip_streams = chunk, ns
        net._ws = None
        with torch.no_grad():
            for i in range(2):
                out = net.render(*rays[i], staged=True, num_steps=96, upsample_steps=96, rng_u=u)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(6):
                out = net.render(*rays[i], staged=True, num_steps=96, upsample_steps=96, rng_u=u)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
        if ref is None: ref = out["image"].clone()
        same = torch.equal(ref, out["image"])
        print(f"chunk {chunk:7d} streams {ns}: {dt*1e3:7.2f} ms/view  {H*W/dt/1e6:6.2f} M rays/s  identical={same}")

"""Headline render time vs hip_ray_chunk (640x480, 96+96, image-ordered)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200)
W, H = 640, 480
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
u = torch.rand(H * W, 96, device=dev)
net.precision = os.environ.get("PREC", "bf16x3")
for pipe, chunk in [(p, c) for p in (True, False) for c in (15360, 20480, 30720, 40960, 51200, 65536, 102400, 153600, 307200)]:
    net.hip_ray_chunk, net.hip_pipeline = chunk, pipe
    with torch.no_grad():
        for _ in range(3):
            net.render(o, d, n, num_steps=96, upsample_steps=96, rng_u=u, image_width=W)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            net.render(o, d, n, num_steps=96, upsample_steps=96, rng_u=u, image_width=W)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"pipeline {pipe} chunk {chunk}: {dt*1e3:.2f} ms/view, {H*W/dt/1e6:.2f} M rays/s")

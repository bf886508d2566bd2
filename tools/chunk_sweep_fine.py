"""cfg2 view time vs chunk size below the Infinity Cache size (does the feature round trip stay on die?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200, deterministic=True)
W, H = 640, 480
poses = _slerp_loop_poses(8, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i + 1], (0.89 * W, 0.89 * W, W / 2, H / 2), H, W) for i in range(8)]
u = torch.rand(H * W, 96, device=dev)
net.precision = "f16x2"
for chunk in [int(c) for c in os.environ.get("CHUNKS", "30720,40960,51200,61440,76800,102400,153600").split(",")]:
    net.hip_ray_chunk = chunk
    net.hip_pipeline_rays = chunk
    with torch.no_grad():
        for i in range(3):
            net.render(*rays[i], num_steps=96, upsample_steps=96, rng_u=u, image_width=W)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(16):
            net.render(*rays[i % 8], num_steps=96, upsample_steps=96, rng_u=u, image_width=W)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
    print(f"chunk {chunk}: {dt*1e3:.2f} ms/view, {H*W/dt/1e6:.2f} M rays/s", flush=True)

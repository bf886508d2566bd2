"""ms per 640x480 view against the number of HIP streams the chunks of a view
alternate over (net.hip_streams), per MLP arithmetic mode.
   python tools/streams_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ucsa_neural_rendering_amd import ops  # noqa: E402
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses  # noqa: E402

dev = torch.device("cuda:0")
net, _ = bench.build_field(dev, train_steps=200)
H, W, T, t = bench.H, bench.W, bench.T_COARSE, bench.T_FINE
intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
poses = _slerp_loop_poses(12, seed=999).to(dev)
rays = [ops.get_rays(poses[i:i + 1], intr, H, W) for i in range(12)]
u = torch.rand(H * W, t, device=dev)
chunks = [int(c) for c in os.environ.get("CHUNKS", "65536").split(",")]
for prec in ("bf16x3", "fp16", "fp32"):
    net.precision = prec
    for chunk in chunks:
        net.hip_ray_chunk = chunk
        for ns in (1, 2, 3):
            net.hip_streams = ns

            def step(i):
                o, d, nrm = rays[i % 12]
                with torch.no_grad():
                    return net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                                      upsample_steps=t, rng_u=u, image_width=W)
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(10):
                step(i)
            torch.cuda.synchronize()
            print(f"{prec} chunk {chunk} streams {ns}: {(time.perf_counter() - t0) * 100:.2f} ms per view", flush=True)

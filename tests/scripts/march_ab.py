"""Deterministic marcher-render timing (fixed parameters, fixed occupancy grid,
fixed rays) for A/B comparisons of kernel variants."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import hip_network_from_oracle, lively_oracle_field, march_scene
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
fld = lively_oracle_field()
net = hip_network_from_oracle(fld, cuda_ray=True).eval()
with torch.no_grad():
    net.sigma_net.params.mul_(float(os.environ.get("SIGMA_GAIN", "0.35")))
_, _, grid, _ = march_scene(4, 5, bound=4.0, H=128, fill=float(os.environ.get("FILL", "0.12")))
net.density_grid = torch.from_numpy(grid).cuda()
net.mean_density = 0.5
W, H = 640, 480
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
with torch.no_grad():
    for _ in range(3):
        out = net.run_cuda(o, d, n, dt_gamma=1 / 256, far_closure=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = net.run_cuda(o, d, n, dt_gamma=1 / 256, far_closure=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"{dt*1e3:.3f} ms/view, {net.last_march_points/(H*W):.1f} pts/ray, rounds {net.last_march_rounds}, checksum {out['image'].double().sum().item():.6f} ws {out['weights_sum'].double().mean().item():.4f}")

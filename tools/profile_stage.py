"""Run the staged forward kernels at bench size a few times (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", "200")))
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
H, W = 480, 640
pose = _slerp_loop_poses(4, seed=999)[:1].to(dev)
o, d, nrm = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 32768
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(N, 96, device=dev, generator=g)
st, rho = bench.stage_times(net, o[0, :N].contiguous(), d[0, :N].contiguous(), nrm[0, :N, 0].contiguous(), u, iters=int(os.environ.get("ITERS", "3")))
print(st, rho)

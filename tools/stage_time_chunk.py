"""composite (k_weights_compact + k_shade16<f16x2>) and sigma-MLP stage times on the bench chunk."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200, deterministic=True)
net.precision = "f16x2"
W, H = 640, 480
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 61440
g = torch.Generator(device=dev).manual_seed(5)
u = torch.rand(N, 96, device=dev, generator=g)
for rep in range(3):
    st, rho = bench.stage_times(net, o[0, :N].contiguous(), d[0, :N].contiguous(), n[0, :N, 0].contiguous(), u, image_width=W, mode="f16x2")
    print(" ".join(f"{k} {v:.3f}" for k, v in st.items() if isinstance(v, float)), f"rho {rho:.3f}", flush=True)

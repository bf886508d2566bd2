"""rocprofv3 target: train the synthetic room (PRE steps; MARCH_TRAIN=1 trains
through the marcher), then render one W x H view ITERS times with the
segmented marcher (run_cuda).  Summarise with TAIL_FRAC to isolate the
renders."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ucsa_neural_rendering_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
pre = int(os.environ.get("PRE", "1500"))
if os.environ.get("MARCH_TRAIN"):
    from tools.march_train import train
    net, ds, _ = train(True, pre, 1 / 256, dev)
    dtg, closure = 1 / 256, False
else:
    net, ds = bench.build_field(dev, train_steps=pre, cuda_ray=True)
    dtg, closure = 1 / 128, True
net.eval()
net.update_extra_state()
caps = tuple(int(c) for c in os.environ.get("CAPS", "32,96,1024").split(","))
W, H = int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses  # noqa: E402
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
torch.cuda.synchronize()
import time
for it in range(int(os.environ.get("ITERS", "10"))):
    t0 = time.perf_counter()
    with torch.no_grad():
        out = net.run_cuda(o, d, n, dt_gamma=dtg, march_caps=caps, far_closure=closure)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("points/ray", net.last_march_points / (H * W), "rounds", net.last_march_rounds, "ms/view", dt * 1e3)

"""rocprofv3 target: pretrain the synthetic room, then render one 320x240 view
ITERS times with the segmented marcher (run_cuda).  PRE / ITERS / CAPS env."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", "1500")),
                            cuda_ray=True)
net.eval()
net.update_extra_state()
caps = tuple(int(c) for c in os.environ.get("CAPS", "32,96,1024").split(","))
item = ds[3]
o, d, n = item["rays_o"][None], item["rays_d"][None], item["direction_norms"][None]
for _ in range(int(os.environ.get("ITERS", "10"))):
    out = net.run_cuda(o, d, n, dt_gamma=1 / 128, march_caps=caps)
torch.cuda.synchronize()
print("points/ray", net.last_march_points / o.shape[1], "rounds", net.last_march_rounds)

"""rocprofv3 target: STEPS marched NeRF training steps on the synthetic room
(4096 rays, dt_gamma 1/256).  Summarise with TAIL_FRAC=0.2 to see the steady
state (air emptied)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tools.march_train import train  # noqa: E402

dev = torch.device("cuda:0")
steps = int(os.environ.get("STEPS", "800"))
net, ds, res = train(True, steps, 1 / 256, dev)
print(res["ms_per_step"], res["points_per_ray_every_50_steps"][-3:])

"""Run a few NeRF train steps (4096 rays x (256+256) samples) for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
log = {}
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", "50")), log=log)
print(log)
from tools.bench_legs.train import train_throughput
r = train_throughput(net, ds, dev, steps=int(os.environ.get("STEPS", "10")),
                           train_precision=os.environ.get("TRAIN_PRECISION", "fp32"))
print(r)

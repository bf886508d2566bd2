"""Per-level time of the hash-grid backward (one launch per level), training sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from ucsa_neural_rendering_amd import ops, _lib
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", "100")))
f = net._field()
aabb = net._aabb_list(True)
N, T = 4096, 256
item = ds[0]
g = torch.Generator(device=dev).manual_seed(3)
inds = torch.randint(0, 240 * 320, (N,), device=dev, generator=g)
o, d = item["rays_o"][inds].contiguous(), item["rays_d"][inds].contiguous()
near, far = ops.near_far_from_aabb(o, d, aabb)
z = ops.sample_coarse(near, far, T, torch.rand(N, T, device=dev, generator=g))
gr = f["grid"]
d_feat = torch.randn(16, N * T, 2, device=dev, generator=g) * 1e-3
grad = torch.zeros_like(net.encoder.params)
tot = 0
for lv in range(16):
    sub = _lib.Grid()
    C.memmove(C.byref(sub), C.byref(gr), C.sizeof(gr))
    sub.n_levels = 1
    sub.level[0] = gr.level[lv]
    df = d_feat[lv:lv + 1].contiguous()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ops.hashgrid_bwd_rays(sub, o, d, z, aabb, df, grad)
    ev0.record()
    for _ in range(5):
        ops.hashgrid_bwd_rays(sub, o, d, z, aabb, df, grad)
    ev1.record(); torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / 5 * 1e3
    tot += us
    print(f"level {lv:2d} res {gr.level[lv].res:5d} scale {gr.level[lv].scale:8.1f} hashed {gr.level[lv].hashed}  {us:8.1f} us")
print("total", tot)

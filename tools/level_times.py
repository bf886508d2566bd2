"""Per-level time of the hash-grid encode kernel (one launch per level)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from ucsa_neural_rendering_amd import ops, _lib
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", "100")))
f = net._field()
aabb = net._aabb_list(False)
H, W = 480, 640
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
pose = _slerp_loop_poses(4, seed=999)[:1].to(dev)
o, d, nrm = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 32768
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
near, far = ops.near_far_from_aabb(o, d, aabb)
z = ops.sample_coarse(near, far, 96)
g = f["grid"]
full = ops.hashgrid_encode_rays(g, f["table"], o, d, z, aabb)
for lv in range(16):
    sub = _lib.Grid()
    C.memmove(C.byref(sub), C.byref(g), C.sizeof(g))
    sub.n_levels = 1
    sub.level[0] = g.level[lv]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.hashgrid_encode_rays(sub, f["table"], o, d, z, aabb)
    ev0.record()
    for _ in range(10):
        out = ops.hashgrid_encode_rays(sub, f["table"], o, d, z, aabb)
    ev1.record(); torch.cuda.synchronize()
    ok = torch.equal(out[0], full[lv])
    print(f"level {lv:2d} res {g.level[lv].res:5d} hashed {g.level[lv].hashed}  {ev0.elapsed_time(ev1)/10*1e3:8.1f} us  match={ok}")

"""Tile-ordered vs ray-ordered hash-grid gather on a full 640x480 view chunk."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", 200)))
f = net._field()
W, H = 640, 480
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
rows = int(os.environ.get("ROWS", 96))
N = rows * W
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
aabb = net._aabb_list(False)
near, far = ops.near_far_from_aabb(o, d, aabb, 0.2)
T = 96
z = ops.sample_coarse(near, far, T, None)
h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb), f["packed_sigma"])
u = torch.rand(N, T, device=dev)
zf = ops.resample(z, sig.view(N, T), u, 1.0)
for name, zz in (("coarse", z), ("fine", zf)):
    a = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb)
    b = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
    print(name, "bit-identical:", bool(torch.equal(a, b)))
    for label, w in (("ray-major", 0), ("tiled", W)):
        for _ in range(3):
            ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=w)
        torch.cuda.synchronize()
        print(f"  {label}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms for {N * T / 1e6:.2f} M samples")

"""x-pair gather (encode_level_hashed) vs plain 8-load gather (encode_level)
on the hashed levels, per kernel and table type (UCSA_ENC_SIMPLE,
UCSA_ENC_SIMPLE_H, UCSA_ENC_SIMPLE_RAYS = "hashed levels below this index use
the plain gather"; csrc/hashgrid.hip simple_gather_below)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200)
f = net._field()
W, H, T = 640, 480, 96
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 96 * W
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
aabb = net._aabb_list(False)
near, far = ops.near_far_from_aabb(o, d, aabb, 0.2)
z = ops.sample_coarse(near, far, T, None)
h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb, image_width=W), f["packed_sigma"])
zf = ops.resample(z, sig.view(N, T), torch.rand(N, T, device=dev), 1.0)
th = ops.table_to_half(f["table"])
# a training batch: 4096 random pixels of a 320x240 frame, tile-ordered, 256 + 256
item = ds[0]
g = torch.Generator(device=dev).manual_seed(7)
inds = ops.tile_order(torch.randint(0, 240 * 320, (4096,), device=dev, generator=g), 320, H=240)
to, td = item["rays_o"][inds].contiguous(), item["rays_d"][inds].contiguous()
tn, tf = ops.near_far_from_aabb(to, td, aabb, 0.2)
tz = ops.sample_coarse(tn, tf, 256, torch.rand(4096, 256, device=dev, generator=g))
th_, ts = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], to, td, tz, aabb), f["packed_sigma"])
tzf = ops.resample(tz, ts.view(4096, 256), torch.rand(4096, 256, device=dev, generator=g), 1.0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


cases = [
    ("tiled fp32 coarse", "UCSA_ENC_SIMPLE", lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb, image_width=W)),
    ("tiled fp32 fine  ", "UCSA_ENC_SIMPLE", lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb, image_width=W)),
    ("tiled half coarse", "UCSA_ENC_SIMPLE_H", lambda: ops.hashgrid_encode_rays(f["grid"], th, o, d, z, aabb, image_width=W)),
    ("tiled half fine  ", "UCSA_ENC_SIMPLE_H", lambda: ops.hashgrid_encode_rays(f["grid"], th, o, d, zf, aabb, image_width=W)),
    ("rays  fp32 train coarse 4096x256", "UCSA_ENC_SIMPLE_RAYS", lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], to, td, tz, aabb)),
    ("rays  fp32 train fine   4096x256", "UCSA_ENC_SIMPLE_RAYS", lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], to, td, tzf, aabb)),
]
for name, env, fn in cases:
    os.environ[env] = "0"
    ref = fn().clone()
    row = []
    for nlev in (0, 8, 12, 16):
        os.environ[env] = str(nlev)
        same = bool(torch.equal(fn(), ref))
        row.append(f"below {nlev:2d}: {timed(fn):.3f} ms{'' if same else ' (DIFFERENT BITS)'}")
    os.environ.pop(env)
    print(f"{name:34s} " + " | ".join(row), flush=True)

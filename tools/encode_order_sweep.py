"""Level-order sweep of k_hashgrid_encode_tiled (UCSA_ENC_ORDER, see
csrc/hashgrid.hip LevelMap) on the bench's 61 440-ray chunk: coarse pass
(linspace depths) and fine pass (resampled depths).  Features are compared
bit for bit with the default order."""
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
N = 96 * W
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
aabb = net._aabb_list(False)
near, far = ops.near_far_from_aabb(o, d, aabb, 0.2)
T = 96
z = ops.sample_coarse(near, far, T, None)
h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb), f["packed_sigma"])
u = torch.rand(N, T, device=dev)
zf = ops.resample(z, sig.view(N, T), u, 1.0)
L = 16
def perm(k, rows):
    return f"{k}:" + ",".join(str(x) for r in rows for x in r)
orders = {
    "level-major, finest first (default)": "",
    "level-major, coarsest first": perm(1, [[l] for l in range(16)]),
    "k=2 (l, 15-l)": perm(2, [[l, 15 - l] for l in range(8)]),
    "k=2 (l, l+8)": perm(2, [[l, l + 8] for l in range(8)]),
    "k=2 (15-l, l) fine first": perm(2, [[15 - l, l] for l in range(8)]),
    "k=4 (l, 15-l, 7-l, 8+l)": perm(4, [[l, 15 - l, 7 - l, 8 + l] for l in range(4)]),
    "k=4 (l, l+4, l+8, l+12)": perm(4, [[l, l + 4, l + 8, l + 12] for l in range(4)]),
    "k=8 (l, l+2, ...)": perm(8, [[l + 2 * i for i in range(8)] for l in range(2)]),
    "k=16 all levels": perm(16, [list(range(16))]),
    "k=2 fine pairs: (0,1)..(14,15)": perm(2, [[2 * l, 2 * l + 1] for l in range(8)]),
}
variants = [("fp32 table", f["table"], lambda zz: ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W))]
for name, zz in (("coarse", z), ("fine", zf)):
    os.environ["UCSA_ENC_ORDER"] = ""
    ops.env_reload()   # the library snapshots its switches once per process
    ref = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W).clone()
    for label, order in orders.items():
        os.environ["UCSA_ENC_ORDER"] = order
        ops.env_reload()   # the library snapshots its switches once per process
        got = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
        same = bool(torch.equal(got, ref))
        for _ in range(3):
            ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
        torch.cuda.synchronize()
        print(f"{name:6s} {label:36s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  bit-identical {same}", flush=True)
os.environ["UCSA_ENC_ORDER"] = ""
ops.env_reload()   # the library snapshots its switches once per process
# hashed levels below index n through the plain 8-load gather (UCSA_ENC_SIMPLE)
for name, zz in (("coarse", z), ("fine", zf)):
    os.environ["UCSA_ENC_SIMPLE"] = "0"
    ops.env_reload()   # the library snapshots its switches once per process
    ref = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W).clone()
    for n in (0, 6, 8, 9, 10, 11, 12, 16):
        os.environ["UCSA_ENC_SIMPLE"] = str(n)
        ops.env_reload()   # the library snapshots its switches once per process
        got = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
        same = bool(torch.equal(got, ref))
        for _ in range(3):
            ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
        torch.cuda.synchronize()
        print(f"{name:6s} simple gather below level {n:2d}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  bit-identical {same}", flush=True)
os.environ["UCSA_ENC_SIMPLE"] = "0"
ops.env_reload()   # the library snapshots its switches once per process

"""Per-LEVEL time of the depth-ordered encoder (k_hashgrid_encode_sorted, one-level
grids built from the real one) on the bench's fine pass, next to the image-ordered
kernel on the same depths, and the several-levels kernel on levels [0, k)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd._lib import Grid
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
zf = ops.resample(z, sig.view(N, T), torch.rand(N, T, device=dev), 1.0)
zs, pix, slot = ops.tile_depth_order(zf, W)
full = f["grid"]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def sub(levels):
    g = Grid()
    g.n_levels, g.n_features, g.total_entries, g.bound = len(levels), full.n_features, full.total_entries, full.bound
    for i, l in enumerate(levels):
        g.level[i] = full.level[l]
    return g


print("level | image-ordered (tiled) | depth-ordered (sorted), per-level kernel   (us, fine pass)")
tot = [0.0, 0.0]
for l in range(full.n_levels):
    g1 = sub([l])
    os.environ["UCSA_ENC_ML"] = "0"
    ops.env_reload()   # the library snapshots its switches once per process
    os.environ["UCSA_ENC_SORTED_ML"] = "0"
    ops.env_reload()   # the library snapshots its switches once per process
    a = 1e3 * timed(lambda: ops.hashgrid_encode_rays(g1, f["table"], o, d, zf, aabb, image_width=W))
    b = 1e3 * timed(lambda: ops.hashgrid_encode_sorted(g1, f["table"], o, d, zs, pix, aabb, T, W))
    tot = [tot[0] + a, tot[1] + b]
    print(f"{l:5d} | {a:7.1f} | {b:7.1f}", flush=True)
print("sum:", [round(x) for x in tot])
for k in (4, 8, 9, 10):
    gk = sub(list(range(k)))
    os.environ["UCSA_ENC_SORTED_ML"] = str(k)
    ops.env_reload()   # the library snapshots its switches once per process
    t = 1e3 * timed(lambda: ops.hashgrid_encode_sorted(gk, f["table"], o, d, zs, pix, aabb, T, W))
    print(f"several-levels kernel, levels 0..{k - 1}: {t:.1f} us ({t / k:.1f} per level)")
for lo in (8, 9, 10):
    gk = sub(list(range(lo, 16)))
    os.environ["UCSA_ENC_SORTED_ML"] = "0"
    ops.env_reload()   # the library snapshots its switches once per process
    t = 1e3 * timed(lambda: ops.hashgrid_encode_sorted(gk, f["table"], o, d, zs, pix, aabb, T, W))
    print(f"per-level kernel, levels {lo}..15 in one launch: {t:.1f} us")

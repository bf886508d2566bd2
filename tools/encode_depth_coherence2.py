"""Round 5 follow-up of encode_depth_coherence.py: why does the real depth-ordered
encoder (0.93 ms) not reach the proxy (0.74 ms)?  Same dealt depth arrays through
both kernels, and variants of the deal:
  zf        the real fine depths
  dealt     pooled per tile, sorted, rank 64 i + p -> (sample i, pixel p)
  dealt-shf the same with the 64 depths of every sample index permuted over the pixels
"""
import os, sys, torch
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
ROWS = 96
N = ROWS * W
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
aabb = net._aabb_list(False)
near, far = ops.near_far_from_aabb(o, d, aabb, 0.2)
T = 96
z = ops.sample_coarse(near, far, T, None)
h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb), f["packed_sigma"])
zf = ops.resample(z, sig.view(N, T), torch.rand(N, T, device=dev), 1.0)


def to_tiles(a):   # [N,T] -> [tiles, 64, T]
    return a.view(ROWS // 8, 8, W // 8, 8, T).permute(0, 2, 1, 3, 4).reshape(-1, 64, T)


def from_tiles(t):
    return t.reshape(ROWS // 8, W // 8, 8, 8, T).permute(0, 2, 1, 3, 4).reshape(N, T).contiguous()


pooled = to_tiles(zf).reshape(-1, 64 * T).sort(dim=1).values.view(-1, T, 64)   # [tiles, i, p]
dealt = from_tiles(pooled.permute(0, 2, 1))
perm = torch.rand(pooled.shape, device=dev).argsort(dim=2)
dealt_shf = from_tiles(torch.gather(pooled, 2, perm).permute(0, 2, 1))


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


for name, zz in (("coarse z", z), ("zf (real fine)", zf), ("dealt", dealt), ("dealt-shf", dealt_shf)):
    t_img = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W))
    zs, pix, slot = ops.tile_depth_order(zz, W)
    t_srt = timed(lambda: ops.hashgrid_encode_sorted(f["grid"], f["table"], o, d, zs, pix, aabb, T, W))
    # how many distinct pixels does a wave (64 consecutive ranks) hold?
    pp = pix.view(-1, 64).long()
    distinct = float((torch.zeros(pp.shape[0], 64, device=dev).scatter_(1, pp, 1.0).sum(1)).mean())
    zspan = float((zs.view(-1, 64).amax(1) - zs.view(-1, 64).amin(1)).mean())
    print(f"{name:16s} image-ordered kernel {t_img:.3f} ms | depth-ordered kernel {t_srt:.3f} ms | "
          f"per wave of the depth order: {distinct:.1f} distinct pixels, z span {zspan:.4f}", flush=True)

"""Follow-up of encode_axis_exp.py: when the table's x axis lies along the VIEW
direction, lanes of a lateral tile share no lines -- do waves of consecutive
samples ALONG a ray (the ray-ordered kernel) do better there?  Coarse and fine
pass, per view: tiled / depth-ordered (shipped) vs ray-ordered."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200)
f = net._field()
H, W, T = bench.H, bench.W, bench.T_COARSE
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
poses = _slerp_loop_poses(23, seed=999).to(dev)
aabb = net._aabb_list(False)
intr = (0.89 * W, 0.89 * W, W / 2, H / 2)


def timed(fn, n=6):
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


for v in range(0, 23, 2):
    o, d, nrm = ops.get_rays(poses[v:v + 1], intr, H, W)
    oo, dd = o[0, 2 * 61440:3 * 61440].contiguous(), d[0, 2 * 61440:3 * 61440].contiguous()
    near, far = ops.near_far_from_aabb(oo, dd, aabb, 0.2)
    z = ops.sample_coarse(near, far, T, None)
    h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, z, aabb), f["packed_sigma"])
    zf = ops.resample(z, sig.view(-1, T), torch.rand(61440, T, device=dev), 1.0)
    zf_sorted = zf.sort(dim=1).values
    tc_t = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, z, aabb, image_width=W))
    tc_r = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, z, aabb))
    zs, pix, slot = ops.tile_depth_order(zf, W)
    tf_s = timed(lambda: ops.hashgrid_encode_sorted(f["grid"], f["table"], oo, dd, zs, pix, aabb, T, W))
    tf_r = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, zf_sorted, aabb))
    tf_t = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, zf, aabb, image_width=W))
    vd = dd[61440 // 2]
    print(f"view {v:2d} dir ({float(vd[0]):+.2f} {float(vd[1]):+.2f} {float(vd[2]):+.2f}) | coarse: tiled {tc_t:.3f}  ray-ordered {tc_r:.3f} | "
          f"fine: depth-ordered {tf_s:.3f}  ray-ordered (z sorted along the ray) {tf_r:.3f}  image-ordered {tf_t:.3f}", flush=True)

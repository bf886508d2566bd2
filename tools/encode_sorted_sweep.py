"""Round 5: the depth-ordered fine pass (csrc/hashgrid_sorted.hip) against the
image-ordered one on the bench's 61 440-ray chunk: sort, encode (by
UCSA_ENC_SORTED_ML / _LEAN) and sigma MLP with the scatter, event-timed; h / sigma
compared bit for bit with the staged pair."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=int(os.environ.get("PRE", 200)))
f = net._field()
f2 = net._field_h2()
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


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, zz in (("coarse", z), ("fine", zf)):
    ref = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W)
    h0, s0 = ops.sigma_mlp_fwd_h2(ref, f2["packed_sigma"])
    t_enc = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zz, aabb, image_width=W))
    t_sig = timed(lambda: ops.sigma_mlp_fwd_h2(ref, f2["packed_sigma"]))
    print(f"{name:6s} image-ordered: encode {t_enc:.3f} ms + sigma(f16x2) {t_sig:.3f} ms = {t_enc + t_sig:.3f} ms")
    zs, pix, slot = ops.tile_depth_order(zz, W)
    t_sort = timed(lambda: ops.tile_depth_order(zz, W))
    for lean in (1, 0):
        os.environ["UCSA_ENC_SORTED_LEAN"] = str(lean)
        for ml in [int(x) for x in os.environ.get("KS", "0,4,8,9,10,12").split(",")]:
            os.environ["UCSA_ENC_SORTED_ML"] = str(ml)
            ops.lib().ucsa_env_reload()      # the library reads its switches once
            got = ops.hashgrid_encode_sorted(f["grid"], f["table"], o, d, zs, pix, aabb, T, W)
            same = bool(torch.equal(got, ref[:, slot.long()]))
            t_e = timed(lambda: ops.hashgrid_encode_sorted(f["grid"], f["table"], o, d, zs, pix, aabb, T, W))
            h1, s1 = ops.sigma_mlp_fwd_scatter(3, got, f2["packed_sigma"], slot)
            same = same and bool(torch.equal(h0, h1) and torch.equal(s0, s1))
            t_s = timed(lambda: ops.sigma_mlp_fwd_scatter(3, got, f2["packed_sigma"], slot))
            print(f"{name:6s} depth-ordered lean={lean} ml={ml:2d}: sort {t_sort:.3f} + encode {t_e:.3f} + sigma/scatter {t_s:.3f} = "
                  f"{t_sort + t_e + t_s:.3f} ms  bit-identical {same}", flush=True)

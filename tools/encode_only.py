"""The hash-grid encoder alone on the bench's 61 440-ray chunk, as
ucsa_render_view runs it (round 5): coarse pass = k_hashgrid_encode_tiled (levels
9-15) + k_hashgrid_encode_tiled_ml (levels 0-8) on the image-ordered samples; fine
pass = k_tile_depth_order2, then k_hashgrid_encode_sorted (9-15) +
k_hashgrid_encode_sorted_ml (0-8) on the depth-ordered samples.  6 launches of
each after the set-up: the program the `--pmc` passes of tools/encode_pmc.sh
profile."""
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
N = 96 * W
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
aabb = net._aabb_list(False)
near, far = ops.near_far_from_aabb(o, d, aabb, 0.2)
T = 96
z = ops.sample_coarse(near, far, T, None)
# (set-up through the ray-ordered kernel, so that every k_hashgrid_encode_tiled
# dispatch of a PASS=coarse / PASS=fine run belongs to that pass; same features)
h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb), f["packed_sigma"])
zf = ops.resample(z, sig.view(N, T), torch.rand(N, T, device=dev), 1.0)
torch.cuda.synchronize()
which = os.environ.get("PASS", "both")
if which in ("both", "coarse"):
    for _ in range(6):
        ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb, image_width=W)
    torch.cuda.synchronize()
if which in ("both", "fine"):
    for _ in range(6):
        zs, pix, slot = ops.tile_depth_order(zf, W)
        ops.hashgrid_encode_sorted(f["grid"], f["table"], o, d, zs, pix, aabb, T, W)
    torch.cuda.synchronize()

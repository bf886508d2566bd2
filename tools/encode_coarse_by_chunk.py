"""Is the COARSE pass depth-divergent on some chunks?  (bench.py's stage_times shows
0.93 ms for the first chunk of its view 0 where tools/encode_only.py's view shows
0.61.)  Per view and 96-row chunk of the bench's views: the image-ordered coarse
encoder, the depth-ordered one (+ its sort), and the far-depth spread inside the
8x8 tiles in units of the sample spacing."""
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


tot = [0.0, 0.0, 0.0]
for v in (0, 3, 7, 11, 15, 19):
    o, d, nrm = ops.get_rays(poses[v:v + 1], intr, H, W)
    for c in range(5):
        oo, dd = o[0, c * 61440:(c + 1) * 61440].contiguous(), d[0, c * 61440:(c + 1) * 61440].contiguous()
        near, far = ops.near_far_from_aabb(oo, dd, aabb, 0.2)
        z = ops.sample_coarse(near, far, T, None)
        t_img = timed(lambda: ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, z, aabb, image_width=W))
        zs, pix, slot = ops.tile_depth_order(z, W)
        t_sort = timed(lambda: ops.tile_depth_order(z, W))
        t_srt = timed(lambda: ops.hashgrid_encode_sorted(f["grid"], f["table"], oo, dd, zs, pix, aabb, T, W))
        ft = far.view(12, 8, 80, 8)
        spread = float(((ft.amax((1, 3)) - ft.amin((1, 3))) / ((far - near).view(12, 8, 80, 8).mean((1, 3)) / T)).mean())
        tot = [tot[0] + t_img, tot[1] + t_srt + t_sort, tot[2] + min(t_img, t_srt + t_sort + 0.05)]
        print(f"view {v:2d} chunk {c}: image-ordered {t_img:.3f} ms | sort {t_sort:.3f} + depth-ordered {t_srt:.3f} | "
              f"far spread inside a tile {spread:5.1f} sample spacings", flush=True)
print(f"sum over 30 chunks: image-ordered {tot[0]:.2f} ms, depth-ordered incl. sort {tot[1]:.2f} ms, "
      f"best of the two per chunk (+0.05 scatter when sorted) {tot[2]:.2f} ms")

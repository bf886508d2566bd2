"""Sample blocks per workgroup of the tiled encoder (UCSA_ENC_SB = 1 / 2 / all;
csrc/hashgrid.hip k_hashgrid_encode_tiled): whole coarse / fine pass on the
bench's 61 440-ray chunk, then per level.
   python tools/encode_sb_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ucsa_neural_rendering_amd import ops  # noqa: E402
from ucsa_neural_rendering_amd._lib import Grid  # noqa: E402
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses  # noqa: E402

dev = torch.device("cuda:0")
net, _ = bench.build_field(dev, train_steps=200)
H, W, T, t = bench.H, bench.W, bench.T_COARSE, bench.T_FINE
o, d, nrm = ops.get_rays(_slerp_loop_poses(4, seed=999)[1:2].to(dev),
                         (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 61440
o, d = o[0, :N].contiguous(), d[0, :N].contiguous()
aabb = net._aabb_list(False)
f = net._field()
near, far = ops.near_far_from_aabb(o, d, aabb)
zc = ops.sample_coarse(near, far, T)
hc, sc = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb, image_width=W), f["packed_sigma"])
zf = ops.resample(zc, sc.view(N, T), torch.rand(N, t, device=dev))
full = f["grid"]
BATCHES = (1, 2, 6)


def timed(fn, n=10):
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


for name, z in (("coarse", zc), ("fine", zf)):
    fn = lambda: ops.hashgrid_encode_rays(full, f["table"], o, d, z, aabb, image_width=W)  # noqa: E731
    os.environ["UCSA_ENC_SB"] = "1"
    ref = fn().clone()
    row = []
    for b in BATCHES:
        os.environ["UCSA_ENC_SB"] = str(b)
        same = bool(torch.equal(fn(), ref))
        row.append(f"sb {b}: {timed(fn):.3f} ms{'' if same else ' (DIFFERENT BITS)'}")
    print(f"{name:7s} pass, fp32 table   " + " | ".join(row), flush=True)

print("level  res  hashed | coarse pass sb 1 / 2 / 6 | fine pass sb 1 / 2 / 6   (us)")
for l in range(full.n_levels):
    g1 = Grid()
    g1.n_levels, g1.n_features, g1.total_entries, g1.bound = 1, full.n_features, full.total_entries, full.bound
    g1.level[0] = full.level[l]
    row = []
    for z in (zc, zf):
        for b in BATCHES:
            os.environ["UCSA_ENC_SB"] = str(b)
            row.append(1e3 * timed(lambda: ops.hashgrid_encode_rays(g1, f["table"], o, d, z, aabb, image_width=W)))
    print(f"{l:5d} {full.level[l].res:5d} {full.level[l].hashed:6d} | "
          + " / ".join(f"{x:6.1f}" for x in row[:3]) + " | " + " / ".join(f"{x:6.1f}" for x in row[3:]), flush=True)
os.environ.pop("UCSA_ENC_SB")

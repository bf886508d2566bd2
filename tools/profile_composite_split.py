"""One process, for `rocprofv3 --kernel-trace --stats -- python3
tools/profile_composite_split.py [fp16]`: the fused composite kernel and the
split pair on the bench's chunk, 10 launches each (kernel names tell them
apart in the trace)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ucsa_neural_rendering_amd import ops  # noqa: E402
from ucsa_neural_rendering_amd._lib import check, lib  # noqa: E402
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses  # noqa: E402

half = "fp16" in sys.argv
dev = torch.device("cuda:0")
net, _ = bench.build_field(dev, train_steps=200)
H, W, T, t = 480, 640, 96, 96
o, d, nrm = ops.get_rays(_slerp_loop_poses(4, seed=999)[1:2].to(dev),
                         (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 61440
o, d, nrm = o[0, :N].contiguous(), d[0, :N].contiguous(), nrm[0, :N, 0].contiguous()
u = torch.rand(N, t, device=dev)
aabb = net._aabb_list(False)
f = net._field_f16() if half else net._field()
sig = ops.sigma_mlp_fwd_f16 if half else ops.sigma_mlp_fwd
near, far = ops.near_far_from_aabb(o, d, aabb)
zc = ops.sample_coarse(near, far, T)
hc, sc = sig(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb, image_width=W),
             f["packed_sigma"])
sc = sc.view(N, T)
zf = ops.resample(zc, sc, u)
hf, sf = sig(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb, image_width=W),
             f["packed_sigma"])
sf = sf.view(N, t)
p = lambda x: C.c_void_p(x.data_ptr())
img, dep, sem = (torch.empty(N, 3, device=dev), torch.empty(N, device=dev),
                 torch.empty(N, 40, device=dev))
for _ in range(10):
    fn = lib().ucsa_composite_fwd_f16 if half else lib().ucsa_composite_fwd
    args = [p(d), p(nrm), p(zc), p(sc), p(hc), p(zf), p(sf), p(hf), p(f["packed_color"]),
            p(f["packed_sem"]), N, T, t, 40, 1.0, p(img), p(dep), p(sem)]
    args += [None, None]
    check(fn(*args, ops._stream()), "fused")
    ops.composite_infer(d, nrm, zc, sc, hc, zf, sf, hf, f["packed_color"], f["packed_sem"],
                        40, half=half)
torch.cuda.synchronize()
print("done")

"""bf16x3 shading (ucsa_composite_infer_x3) on the bench's chunk: time per
launch shape, and distance to the f32-input MFMA kernel / the f16 one.
   python tools/x3_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ucsa_neural_rendering_amd import ops  # noqa: E402
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses  # noqa: E402

dev = torch.device("cuda:0")
net, _ = bench.build_field(dev, train_steps=200)
H, W, T, t = 480, 640, 96, 96
o, d, nrm = ops.get_rays(_slerp_loop_poses(4, seed=999)[1:2].to(dev),
                         (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 61440
o, d, nrm = o[0, :N].contiguous(), d[0, :N].contiguous(), nrm[0, :N, 0].contiguous()
u = torch.rand(N, t, device=dev)
aabb = net._aabb_list(False)
f = net._field()
fh = net._field_f16()
near, far = ops.near_far_from_aabb(o, d, aabb)
zc = ops.sample_coarse(near, far, T)
hc, sc = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb, image_width=W), f["packed_sigma"])
sc = sc.view(N, T)
zf = ops.resample(zc, sc, u)
hf, sf = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb, image_width=W), f["packed_sigma"])
sf = sf.view(N, t)
pc3 = ops.mlp_pack_x3(1, net.color_net.params)
ps3 = ops.mlp_pack_x3(2, net.semantics_net.params, 40)


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


ref = ops.composite_fwd(d, nrm, zc, sc, hc, zf, sf, hf, f["packed_color"], f["packed_sem"], 40)[:3]
print("fp32 fused: %.3f ms" % timed(lambda: ops.composite_fwd(d, nrm, zc, sc, hc, zf, sf, hf, f["packed_color"], f["packed_sem"], 40)), flush=True)
h16 = ops.composite_infer(d, nrm, zc, sc, hc, zf, sf, hf, fh["packed_color"], fh["packed_sem"], 40, half=True)
print("f16 split: %.3f ms" % timed(lambda: ops.composite_infer(d, nrm, zc, sc, hc, zf, sf, hf, fh["packed_color"], fh["packed_sem"], 40, half=True)), flush=True)
for v in ("0", "1", "2", "3"):
    os.environ["UCSA_SHADE_VARIANT"] = v
    ops.env_reload()   # the library snapshots its switches once per process
    fn = lambda: ops.composite_infer(d, nrm, zc, sc, hc, zf, sf, hf, pc3, ps3, 40, x3=True)
    out = fn()
    torch.cuda.synchronize()
    ms = timed(fn)
    print("x3 variant %s: %.3f ms; max |x3 - fp32| image %.3e depth %.3e sem %.3e ; f16: image %.3e sem %.3e" % (
        v, ms, (out[0] - ref[0]).abs().max(), (out[1] - ref[1]).abs().max(), (out[2] - ref[2]).abs().max(),
        (h16[0] - ref[0]).abs().max(), (h16[2] - ref[2]).abs().max()), flush=True)
for v in ("0", "1", "2", "3"):
    os.environ["UCSA_SHADE_VARIANT"] = v
    ops.env_reload()   # the library snapshots its switches once per process
    fn = lambda: ops.composite_infer(d, nrm, zc, sc, hc, zf, sf, hf, fh["packed_color"], fh["packed_sem"], 40, half=True)
    out = fn()
    torch.cuda.synchronize()
    print("f16 variant %s: %.3f ms; identical to variant 0: %s" % (
        v, timed(fn), torch.equal(out[0], h16[0]) and torch.equal(out[2], h16[2])), flush=True)
os.environ.pop("UCSA_SHADE_VARIANT")
ops.env_reload()   # the library snapshots its switches once per process

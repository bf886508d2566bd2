"""Round 5: how many leading levels go through k_hashgrid_encode_tiled_ml (several
levels per workgroup, csrc/hashgrid.hip; UCSA_ENC_ML) on the bench's 61 440-ray
chunk -- coarse pass (linspace depths) and fine pass (resampled depths), fp32
table.  Features are compared bit for bit with UCSA_ENC_ML=0 (the round-3/4
per-level kernel on every level)."""
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
os.environ["UCSA_ENC_ML"] = "0"
ops.env_reload()   # the library snapshots its switches once per process
h, sig = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, z, aabb), f["packed_sigma"])
zf = ops.resample(z, sig.view(N, T), torch.rand(N, T, device=dev), 1.0)
th = net._table_half()


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


ks = [int(x) for x in os.environ.get("KS", "0,2,4,6,7,8,9,10,11,12,16").split(",")]
for tname, tab in (("fp32 table", f["table"]), ("fp16 table", th)):
    enc = (lambda zz: ops.hashgrid_encode_rays(f["grid"], tab, o, d, zz, aabb, image_width=W))
    for name, zz in (("coarse", z), ("fine", zf)):
        os.environ["UCSA_ENC_ML"] = "0"
        ops.env_reload()   # the library snapshots its switches once per process
        ref = enc(zz).clone()
        row = []
        for k in ks:
            os.environ["UCSA_ENC_ML"] = str(k)
            ops.env_reload()   # the library snapshots its switches once per process
            same = bool(torch.equal(enc(zz), ref))
            row.append(f"{k}: {timed(lambda: enc(zz)):.3f}{'' if same else ' MISMATCH'}")
        print(f"{tname} {name:6s} ms by UCSA_ENC_ML | " + " | ".join(row), flush=True)

"""Why does stage_times report the coarse-pass encoder at ~0.93 ms when the same
launches alone take 0.61 ms?  The coarse encode timed back to back and after other
work (a whole chunk's render, a 755 MB memset, an idle gap)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tools.bench_legs.render_modes import stage_times
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200)
net.hip_ray_chunk = 65536
net.precision = "f16x2"
H, W = bench.H, bench.W
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
o, d, nrm = ops.get_rays(_slerp_loop_poses(4, seed=999)[1:2].to(dev), (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
chunk = 61440
u = torch.rand(H * W, bench.T_FINE, device=dev)
cin = (o[0, :chunk].contiguous(), d[0, :chunk].contiguous(), nrm[0, :chunk, 0].contiguous(), u[:chunk])
st, rho = stage_times(net, *cin, image_width=W, mode="f16x2")
print("stage_times:", {k: round(v, 3) for k, v in st.items()})
f = net._field_h2()
aabb = net._aabb_list(False)
oo, dd = cin[0], cin[1]
near, far = ops.near_far_from_aabb(oo, dd, aabb)
zc = ops.sample_coarse(near, far, 96)
ev = lambda: torch.cuda.Event(enable_timing=True)


def enc():
    return ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, zc, aabb, image_width=W)


def whole_chunk():
    with torch.no_grad():
        net.render(o[:, :chunk], d[:, :chunk], nrm[:, :chunk], staged=True, perturb=False,
                   num_steps=96, upsample_steps=96, rng_u=u[:chunk], image_width=W)


for name, pre in (("back to back", lambda: None),
                  ("after a whole chunk's render", whole_chunk),
                  ("after a 755 MB memset", lambda: torch.zeros(16, chunk * 96, 2, device=dev)),
                  ("after 30 ms idle", lambda: (torch.cuda.synchronize(), time.sleep(0.03)))):
    ts = []
    for _ in range(6):
        pre()
        a, b, c = ev(), ev(), ev()
        a.record(); x = enc(); b.record(); y = enc(); c.record()
        torch.cuda.synchronize()
        ts.append((a.elapsed_time(b), b.elapsed_time(c)))
    print(f"{name:40s}: first {sum(t[0] for t in ts[1:]) / 5:.3f} ms, second right after it {sum(t[1] for t in ts[1:]) / 5:.3f} ms")

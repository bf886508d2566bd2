"""Train the ORACLE field on the CPU the way bench.build_field trains the HIP one
(200 Adam steps, 4096 rays x (96+96), the synthetic room; ~6 s per step on 8
cores) -- the field tests/scripts/oracle_self_noise.py renders.  No GPU.
   python tests/scripts/train_oracle_field_cpu.py [steps] [out.pt]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import field as ofield, losses as olosses, renderer as oren
from oracle.rays import pixel_rays
from tests.util import AABB4
from ucsa_neural_rendering_amd.dataset.synthetic_scene import SyntheticRoom, _slerp_loop_poses
torch.set_num_threads(8)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
OUT = sys.argv[2] if len(sys.argv) > 2 else "/tmp/oracle_field.pt"
C, H, W, T, t = 40, 240, 320, 96, 96
room = SyntheticRoom(0, n_classes=C)
poses = _slerp_loop_poses(16, seed=123)
intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
views = []
for i in range(16):
    o, d, n = pixel_rays(poses[i:i + 1], intr, H, W)
    t_hit, rgb, label = room.cast(o[0], d[0])
    views.append(dict(o=o[0], d=d[0], n=n[0].reshape(-1), rgb=rgb, label=label, depth=(t_hit / n[0].reshape(-1)).half().float()))
fld = ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=123)
fld.requires_grad_(True)
st = [dict(m=torch.zeros_like(p), v=torch.zeros_like(p)) for p in fld.parameters()]
g = torch.Generator().manual_seed(123)
t0 = time.time()
for k in range(STEPS):
    v = views[k % 16]
    i = torch.randint(0, H * W, (4096,), generator=g)
    out = oren.run(fld, v["o"][i][None], v["d"][i][None], v["n"][i][None], AABB4, num_steps=T, upsample_steps=t,
                   t_rand=torch.rand(4096, T, generator=g), u=torch.rand(4096, t, generator=g))
    lc, ls, ld = olosses.nerf_losses(out["image"], out["semantics"], out["depth"], v["rgb"][i][None], v["label"][i][None], v["depth"][i][None], 1.0)
    loss = olosses.nerf_total_loss(lc, ls, ld)
    for p in fld.parameters():
        p.grad = None
    loss.backward()
    with torch.no_grad():
        for j, (p, s) in enumerate(zip(fld.parameters(), st)):
            pn, s["m"], s["v"] = olosses.adam_step(p, p.grad, s["m"], s["v"], k + 1, 1e-2, weight_decay=0.0 if j == 0 else 1e-6)
            p.copy_(pn)
    if k % 10 == 0 or k == STEPS - 1:
        print(k, float(loss), f"{time.time() - t0:.0f}s", flush=True)
torch.save({"grid": fld.grid_params.detach(), "sigma": fld.sigma_params.detach(), "color": fld.color_params.detach(), "sem": fld.sem_params.detach()}, OUT)
print("saved")

"""Diagnostic: gradient of the NeRF loss at the tcnn-style INITIAL state
(grid U(-1e-4, 1e-4)), HIP `train_precision: tcnn` (and fp32) under a loss
scale vs the fp16-emulating (and fp32) oracle -- relative L2 and cosine per
parameter tensor.  python tests/scripts/tcnn_init_grad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import field as ofield, losses as olosses, renderer as oren  # noqa: E402
from tests.util import AABB4, hip_network_from_oracle, make_rays  # noqa: E402
from ucsa_neural_rendering_amd import losses as ul  # noqa: E402

torch.set_num_threads(16)
N, T, t, C = 512, 32, 32, 40
o, d, nrm = make_rays(N, 3)
g = torch.Generator().manual_seed(0)
rt, ru = torch.rand(N, T, generator=g), torch.rand(N, t, generator=g)
rgb, lab, dep = torch.rand(1, N, 3, generator=g), torch.randint(0, C, (1, N), generator=g), torch.rand(1, N, generator=g) * 3 + 0.5


def oracle_grads(tcnn):
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=123)
    fld.emulate_fp16, fld.fp16_table = tcnn, tcnn
    fld.requires_grad_(True)
    out = oren.run(fld, o[None], d[None], nrm[None], AABB4, num_steps=T, upsample_steps=t, t_rand=rt, u=ru)
    lc, ls, ld = olosses.nerf_losses(out["image"], out["semantics"], out["depth"], rgb, lab, dep, 1.0)
    olosses.nerf_total_loss(lc, ls, ld).backward()
    return [p.grad.clone() for p in fld.parameters()]


def hip_grads(prec, scale):
    net = hip_network_from_oracle(ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=123)).train()
    net.train_precision = prec
    out = net.render(o[None].cuda(), d[None].cuda(), nrm[None].cuda(), perturb=True, num_steps=T,
                     upsample_steps=t, rng_t=rt.cuda(), rng_u=ru.cuda())
    lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], rgb.cuda(), lab.cuda(), dep.cuda(), 1.0)
    (ul.nerf_total_loss(lc, ls, ld) * scale).backward()
    ps = [net.encoder.params, net.sigma_net.params, net.color_net.params, net.semantics_net.params]
    return [(p.grad / scale).cpu() for p in ps]


def cmp(tag, a, b):
    for name, x, y in zip(("grid", "sigma", "color", "sem"), a, b):
        x, y = x.double().reshape(-1), y.double().reshape(-1)
        rel = float((x - y).norm() / y.norm().clamp_min(1e-30))
        cos = float(torch.dot(x, y) / (x.norm() * y.norm()).clamp_min(1e-300))
        nz = (int((x != 0).sum()), int((y != 0).sum()))
        print(f"{tag} {name}: rel L2 {rel:.3e} cos {cos:.6f} |g| {float(y.norm()):.3e} nonzeros hip/oracle {nz}")


o32, o16 = oracle_grads(False), oracle_grads(True)
cmp("oracle tcnn-emulation vs oracle fp32", o16, o32)
for scale in (1.0, 65536.0):
    cmp(f"hip fp32 (scale {scale:g}) vs oracle fp32", hip_grads("fp32", scale), o32)
    cmp(f"hip tcnn (scale {scale:g}) vs oracle tcnn", hip_grads("tcnn", scale), o16)
    cmp(f"hip tcnn (scale {scale:g}) vs oracle fp32", hip_grads("tcnn", scale), o32)

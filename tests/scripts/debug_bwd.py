"""Debug aid: gradients of the HIP render vs the oracle on a tiny batch (checker script, lives under tests/ because it imports the oracle)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import renderer as oren
from tests.util import AABB4, hip_network_from_oracle, lively_oracle_field, make_rays
N,T,t,perturb = [int(x) for x in sys.argv[1:5]] if len(sys.argv)>4 else (48,16,16,1)
fld = lively_oracle_field().requires_grad_(True)
net = hip_network_from_oracle(fld).train()
o, d, norms = make_rays(N, 300 + N)
g = torch.Generator().manual_seed(N)
t_rand = torch.rand(N, T, generator=g) if perturb else None
u = torch.rand(N, max(t, 1), generator=g)[:, :t]
ci = torch.rand(1, N, 3, generator=g); cd = torch.rand(1, N, generator=g); cs = torch.rand(1, N, 40, generator=g)
which = sys.argv[5] if len(sys.argv)>5 else "all"
def L(r, dev):
    l = 0
    if which in ("all","img"): l = l + (r["image"] * ci.to(dev)).sum()
    if which in ("all","dep"): l = l + (r["depth"] * cd.to(dev)).sum()
    if which in ("all","sem"): l = l + (r["semantics"] * cs.to(dev)).sum()
    return l
ref = oren.run(fld, o[None], d[None], norms[None], AABB4, num_steps=T, upsample_steps=t, t_rand=t_rand, u=u if t else None, return_aux=True)
L(ref,"cpu").backward()
res = net.render(o[None].cuda(), d[None].cuda(), norms[None].cuda(), perturb=bool(perturb), num_steps=T, upsample_steps=t,
                 rng_t=None if t_rand is None else t_rand.cuda(), rng_u=u.cuda() if t else None)
L(res,"cuda").backward()
def rel(a,b):
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    return float((a-b).abs().max()/b.abs().max().clamp_min(1e-30))
print("mask count", int(ref["aux"]["mask"].sum()), "near-threshold", int(((ref["aux"]["weights"]-1e-4).abs()<2e-6).sum()))
gs, rs = net.semantics_net.params.grad, fld.sem_params.grad
print("sem W1", rel(gs[:1024], rs[:1024]), "W2", rel(gs[1024:], rs[1024:]))
w1g = gs[:1024].view(64,16).cpu(); w1r = rs[:1024].view(64,16)
print("sem W1 col err", (w1g-w1r).abs().max(0)[0])
print("sem W1 col ref", w1r.abs().max(0)[0])
print("sem W1 row err", (w1g-w1r).abs().max(1)[0])
gc, rc = net.color_net.params.grad, fld.color_params.grad
if gc is None: sys.exit(0)
print("col W1", rel(gc[:2048], rc[:2048]), "W2", rel(gc[2048:6144], rc[2048:6144]), "W3", rel(gc[6144:], rc[6144:]))
gg, rg = net.sigma_net.params.grad, fld.sigma_params.grad
print("sig W1", rel(gg[:2048], rg[:2048]), "W2", rel(gg[2048:], rg[2048:]))
print("grid", rel(net.encoder.params.grad, fld.grid_params.grad))
w2g = gs[1024:].view(48,64).cpu(); w2r = rs[1024:].view(48,64)
print("sem W2 row err", (w2g-w2r).abs().max(1)[0][:48])
print("sem W2 row ref max", w2r.abs().max(1)[0][:48])

"""Oracle (test infrastructure): fp32 CPU restatement of the volume renderer.

Follows reference ``nr4seg/nerf/renderer_semantics.py``: ``sample_pdf`` :10-46,
``SemanticNeRFRenderer.run`` :123-299 and the chunked ``render`` :301-358.

The reference draws two random tensors inside ``run`` -- ``t_rand [N,T]`` when
``perturb`` (:166) and ``u [N,t]`` inside ``sample_pdf`` (:28, always random
because ``det=False`` is hard-coded at :206).  Here both cross the interface as
explicit tensors so results can be compared value-for-value; the pinning test
(tests/test_golden_renderer.py) replays the same tensors into the reference.
"""
from __future__ import annotations

from typing import Optional

import torch

from .rays import near_far_from_aabb


def coarse_z(nears, fars, T: int, t_rand: Optional[torch.Tensor]):
    """:154-168. nears/fars [N,1] -> z [N,T] (stratified if t_rand given)."""
    lin = torch.linspace(0.0, 1.0, T).unsqueeze(0)
    z = nears + (fars - nears) * lin.expand(nears.shape[0], T)
    if t_rand is not None:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], dim=-1)
        lower = torch.cat([z[:, :1], mids], dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


def positions(rays_o, rays_d, z, aabb):
    """:171-173 -- o + d*z, clipped to the box."""
    p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
    return torch.min(torch.max(p, aabb[:3]), aabb[3:])


def alpha_weights(z, sigma, density_scale: float):
    """:185-198 / :238-247 -- last interval is 1e10 wide; transmittance uses
    (1 - alpha + 1e-15)."""
    deltas = z[:, 1:] - z[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
    alphas = 1 - torch.exp(-deltas * density_scale * sigma)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-15],
                        dim=-1)
    weights = alphas * torch.cumprod(shifted, dim=-1)[:, :-1]
    return deltas, weights


def inverse_cdf(bins, weights, u):
    """:10-46 with ``u`` supplied.  bins [N,T-1], weights [N,T-2], u [N,t]."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(hi - 1, min=0)
    hi = torch.clamp(hi, max=cdf.shape[-1] - 1)
    c0 = torch.gather(cdf, 1, lo)
    c1 = torch.gather(cdf, 1, hi)
    b0 = torch.gather(bins, 1, lo)
    b1 = torch.gather(bins, 1, hi)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


def run(field,
        rays_o,
        rays_d,
        direction_norms,
        aabb,
        num_steps: int = 256,
        upsample_steps: int = 256,
        density_scale: float = 1.0,
        t_rand: Optional[torch.Tensor] = None,
        u: Optional[torch.Tensor] = None,
        min_near: float = 0.2,
        return_aux: bool = False):
    """One un-chunked pass.  ``t_rand`` None <=> perturb=False.
    ``u`` [N, upsample_steps] is required when upsample_steps > 0."""
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    norms = direction_norms.contiguous().view(-1)
    N = rays_o.shape[0]
    C = field.C

    nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    nears = nears.unsqueeze(-1)
    fars = fars.unsqueeze(-1)

    z = coarse_z(nears, fars, num_steps, t_rand)
    xyz = positions(rays_o, rays_d, z, aabb)
    den = field.density(xyz.reshape(-1, 3))
    sigma = den["sigma"].view(N, num_steps)
    geo = den["geo_feat"].view(N, num_steps, -1)

    aux = {}
    if upsample_steps > 0:
        assert u is not None and u.shape == (N, upsample_steps)
        with torch.no_grad():
            deltas, w = alpha_weights(z, sigma, density_scale)
            z_mid = z[:, :-1] + 0.5 * deltas[:, :-1]
            new_z = inverse_cdf(z_mid, w[:, 1:-1], u).detach()
            new_xyz = positions(rays_o, rays_d, new_z, aabb)
        den2 = field.density(new_xyz.reshape(-1, 3))
        sigma2 = den2["sigma"].view(N, upsample_steps)
        geo2 = den2["geo_feat"].view(N, upsample_steps, -1)

        z = torch.cat([z, new_z], dim=1)
        z, order = torch.sort(z, dim=1)
        xyz = torch.gather(torch.cat([xyz, new_xyz], dim=1), 1,
                           order.unsqueeze(-1).expand(-1, -1, 3))
        sigma = torch.gather(torch.cat([sigma, sigma2], dim=1), 1, order)
        geo_all = torch.cat([geo, geo2], dim=1)
        geo = torch.gather(geo_all, 1,
                           order.unsqueeze(-1).expand_as(geo_all))
        aux["new_z"] = new_z
        aux["order"] = order
        aux["z_mid_coarse"] = z_mid      # bins / weights / u of the inverse CDF:
        aux["w_coarse"] = w              # tests use them to tell which fine
        aux["u"] = u                     # samples sit on sample_pdf's denom step

    _, weights = alpha_weights(z, sigma, density_scale)
    mask = weights > 1e-4  # same mask for colour and semantics (:249-250)

    S = z.shape[1]
    dirs = rays_d.view(-1, 1, 3).expand(N, S, 3)
    flat_geo = geo.reshape(N * S, -1)
    rgbs = field.color(xyz.reshape(-1, 3), dirs.reshape(-1, 3),
                       mask=mask.reshape(-1), geo_feat=flat_geo).view(N, S, 3)
    probs = field.semantics(xyz.reshape(-1, 3), dirs.reshape(-1, 3),
                            mask=mask.reshape(-1),
                            geo_feat=flat_geo).view(N, S, C)

    # semantic weights are detached (:270); colour/depth weights keep grad but
    # are zeroed in place outside the mask (:271)
    w_sem = torch.where(mask, weights.detach(), torch.zeros_like(weights))
    w_rgb = torch.where(mask, weights, torch.zeros_like(weights))

    depth = torch.sum(w_rgb * z, dim=-1) / norms
    image = torch.sum(w_rgb.unsqueeze(-1) * rgbs, dim=-2)
    semantics = torch.sum(w_sem.unsqueeze(-1) * probs, dim=-2)

    out = {
        "depth": depth.view(*prefix),
        "image": image.view(*prefix, 3),
        "semantics": semantics.view(*prefix, C),
    }
    if return_aux:
        aux.update(z=z, sigma=sigma, weights=weights, mask=mask, rgbs=rgbs,
                   probs=probs, nears=nears, fars=fars)
        out["aux"] = aux
    return out


def render(field,
           rays_o,
           rays_d,
           direction_norms,
           aabb,
           staged: bool = False,
           max_ray_batch: int = 4096,
           t_rand: Optional[torch.Tensor] = None,
           u: Optional[torch.Tensor] = None,
           **kw):
    """:301-358.  rays [B,N,3]; ``u`` [B,N,t], ``t_rand`` [B,N,T] or None.
    Chunking slices the random tensors along with the rays, which is exactly
    what replaying one recorded tensor per chunk does in the pinning test."""
    B, N = rays_o.shape[:2]
    if not staged:
        return run(field, rays_o, rays_d, direction_norms, aabb,
                   t_rand=None if t_rand is None else t_rand.reshape(B * N, -1),
                   u=None if u is None else u.reshape(B * N, -1), **kw)
    depth = torch.empty(B, N)
    image = torch.empty(B, N, 3)
    sem = torch.empty(B, N, field.C)
    for b in range(B):
        for head in range(0, N, max_ray_batch):
            tail = min(head + max_ray_batch, N)
            r = run(field, rays_o[b:b + 1, head:tail],
                    rays_d[b:b + 1, head:tail],
                    direction_norms[b:b + 1, head:tail], aabb,
                    t_rand=None if t_rand is None else t_rand[b, head:tail],
                    u=None if u is None else u[b, head:tail], **kw)
            depth[b:b + 1, head:tail] = r["depth"]
            image[b:b + 1, head:tail] = r["image"]
            sem[b:b + 1, head:tail] = r["semantics"]
    return {"depth": depth, "image": image, "semantics": sem}

"""Oracle (test infrastructure): seeded fields, rays and marching scenes
shared by ``tests/``, ``__graft_entry__.smoke()`` and the measurement helpers
under ``tests/scripts``.  Nothing under ``ucsa_neural_rendering_amd/`` imports
this module."""
import numpy as np
import torch

from . import field as ofield

AABB4 = torch.tensor([-4.0, -4, -4, 4, 4, 4])


def make_rays(n, seed, inside=True):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(n, 3, generator=g) * 2 - 1) * (2.5 if inside else 6.0)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    norms = 1.0 + torch.rand(n, 1, generator=g) * 0.3
    return o, d, norms


def lively_oracle_field(C=40, grid_seed=77, grid_amp=3.0, seed=123):
    """The field used by the G5 fixtures: seeded MLPs, grid ~ U(-amp, amp)."""
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=seed)
    gs = torch.Generator().manual_seed(grid_seed)
    fld.grid_params = (torch.rand(fld.grid.n_params, generator=gs) * 2 - 1) * grid_amp
    return fld


def hip_network_from_oracle(fld, device="cuda", cuda_ray=False):
    """A HIP SemanticNeRFNetwork carrying exactly the oracle's parameters."""
    from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
        SemanticNeRFNetwork
    net = SemanticNeRFNetwork(encoding="hashgrid", bound=fld.bound,
                              cuda_ray=cuda_ray, density_scale=1,
                              num_semantic_classes=fld.C)
    with torch.no_grad():
        net.encoder.params.copy_(fld.grid_params)
        net.sigma_net.params.copy_(fld.sigma_params)
        net.color_net.params.copy_(fld.color_params)
        net.semantics_net.params.copy_(fld.sem_params)
    return net.to(device)


def maxabs(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


# ---- occupancy-grid marching scenes (numpy; shared by CPU and GPU tests) ----
def march_scene(N, seed, bound=2.0, H=32, fill=0.35, outside=True):
    """Rays aimed into the box plus a blobby cascade grid [C,H,H,H] with about
    `fill` of the cells above the density threshold."""
    import math
    rs = np.random.RandomState(seed)
    C = max(1, 1 + math.ceil(math.log2(bound)))
    lo = 1.6 if outside else 0.6
    o = ((rs.rand(N, 3) * 2 - 1) * bound * lo).astype(np.float32)
    tgt = ((rs.rand(N, 3) * 2 - 1) * bound * 0.5).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    # low-frequency noise thresholded -> connected occupied blobs
    coarse = rs.rand(C, H // 4, H // 4, H // 4).astype(np.float32)
    grid = np.repeat(np.repeat(np.repeat(coarse, 4, 1), 4, 2), 4, 3)
    grid = np.where(grid < fill, grid + 0.5, grid * 0.001).astype(np.float32)
    return o, d, grid, C


def slab_near_far(o, d, bound, min_near=0.2):
    """numpy slab test with the semantics of oracle.rays.near_far_from_aabb."""
    from . import rays as orays
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)
    n, f = orays.near_far_from_aabb(torch.from_numpy(o), torch.from_numpy(d),
                                    aabb, min_near)
    return n.numpy(), f.numpy()

"""Mirror of reference ``nr4seg/nerf/raymarching/raymarching.py``.

Only ``near_far_from_aabb`` is live in the reference
(renderer_semantics.py:150) -- it is backed here by the HIP kernel
``ucsa_near_far_from_aabb`` instead of the JIT-built CUDA extension.  The
occupancy-grid marching functions are dormant in the reference and are listed
in SURVEY 8f as "next"; calling them raises.
"""
import torch

from ... import ops


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """reference raymarching.py:12-47: rays [N,3] fp32 -> nears, fars [N]."""
    if not rays_o.is_cuda:
        rays_o = rays_o.cuda()
    if not rays_d.is_cuda:
        rays_d = rays_d.cuda()
    with torch.no_grad():
        return ops.near_far_from_aabb(rays_o.float(), rays_d.float(), aabb,
                                      min_near)


def _dormant(name):

    def f(*a, **k):
        raise NotImplementedError(
            f"raymarching.{name} is dormant in the reference (cuda_ray=False "
            "is hard-coded, joint_train_lightning_net.py:29-35) and is not "
            "part of this build yet (SURVEY 8f rank 1)")

    f.__name__ = name
    return f


march_rays_train = _dormant("march_rays_train")
composite_rays_train = _dormant("composite_rays_train")
march_rays = _dormant("march_rays")
composite_rays = _dormant("composite_rays")
compact_rays = _dormant("compact_rays")

"""Mirror of reference ``nr4seg/nerf/raymarching/raymarching.py``.

Same function names, argument order and return values; every function is
backed by a HIP kernel of ``libucsa_hip.so`` instead of the JIT-built CUDA
extension ``_raymarching`` (reference backend.py:45-55).  Only
``near_far_from_aabb`` is live in the reference (renderer_semantics.py:150);
the occupancy-grid marching functions are dormant there (``cuda_ray=False``,
joint_train_lightning_net.py:29-35) and are SURVEY 8f rank 1 here.

Differences, all deliberate:
  * output spans (``rays[:, 1]``) and compacted slots come from prefix sums
    over the ray index, so they are deterministic and in ray order; the CUDA
    kernels hand them out with ``atomicAdd`` in arrival order;
  * ``composite_rays`` and the two ``*_semantics`` functions are callable (the
    reference declares them in Python, raymarching.py:249-360,455-558, but
    does not bind them, bindings.cpp:12-16).  Semantic channels are composited
    with detached weights, as on the live path (renderer_semantics.py:268-271);
  * there is no CPU fallback: CPU tensors are moved to the GPU exactly where
    the reference does so, everything else must already be there.
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


def _cuda(t):
    return t if t.is_cuda else t.cuda()


# ----------------------------------------
# train functions
# ----------------------------------------


def march_rays_train(rays_o, rays_d, bound, density_grid, mean_density, nears,
                     fars, step_counter=None, mean_count=-1, perturb=False,
                     align=-1, force_all_rays=False, dt_gamma=0):
    """reference raymarching.py:54-166 (forward only).

    Returns xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] int32
    (ray id, first point, point count)."""
    with torch.no_grad():
        rays_o = _cuda(rays_o).float().contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).float().contiguous().view(-1, 3)
        density_grid = _cuda(density_grid).float().contiguous()
        N = rays_o.shape[0]
        dev = rays_o.device
        M = N * 1024
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        xyzs = torch.zeros(M, 3, device=dev)
        dirs = torch.zeros(M, 3, device=dev)
        deltas = torch.zeros(M, 2, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        ops.march_rays_train(rays_o, rays_d, density_grid, float(mean_density),
                             float(bound), float(dt_gamma), M, nears, fars,
                             xyzs, dirs, deltas, rays, step_counter,
                             int(perturb))
        if force_all_rays or mean_count <= 0:
            m = int(step_counter[0].item())  # D2H copy, as in the reference
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


class _composite_rays_train(torch.autograd.Function):
    """reference raymarching.py:169-246."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays):
        sigmas = sigmas.float().contiguous()
        rgbs = rgbs.float().contiguous()
        deltas = deltas.float().contiguous()
        ws, depth, image, _ = ops.composite_rays_train_fwd(sigmas, rgbs, None,
                                                           deltas, rays)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, ws, image)
        return ws, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is not propagated (reference :209)
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        g_sig, g_rgb, _ = ops.composite_rays_train_bwd(
            grad_weights_sum.contiguous(), grad_image.contiguous(), None,
            sigmas, rgbs, deltas, rays, ws, image)
        return g_sig, g_rgb, None, None


composite_rays_train = _composite_rays_train.apply


class _composite_rays_train_semantics(torch.autograd.Function):
    """reference raymarching.py:249-360."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, local_semantics, deltas, rays,
                num_semantics_classes):
        sigmas = sigmas.float().contiguous()
        rgbs = rgbs.float().contiguous()
        local_semantics = local_semantics.float().contiguous()
        deltas = deltas.float().contiguous()
        if local_semantics.shape[-1] != num_semantics_classes:
            raise ValueError("local_semantics must be [M, num_semantics_classes]")
        ws, depth, image, sem = ops.composite_rays_train_fwd(
            sigmas, rgbs, local_semantics, deltas, rays)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, ws, image)
        return ws, depth, image, sem

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image, grad_semantics):
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        g_sig, g_rgb, g_ls = ops.composite_rays_train_bwd(
            grad_weights_sum.contiguous(), grad_image.contiguous(),
            grad_semantics.contiguous(), sigmas, rgbs, deltas, rays, ws, image)
        return g_sig, g_rgb, g_ls, None, None, None


composite_rays_train_semantics = _composite_rays_train_semantics.apply

# ----------------------------------------
# infer functions
# ----------------------------------------


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
               density_grid, mean_density, near, far, align=-1, perturb=False,
               dt_gamma=0):
    """reference raymarching.py:367-452 -> xyzs, dirs [M,3], deltas [M,2] with
    M = n_alive * n_step (padded to `align`)."""
    with torch.no_grad():
        rays_o = _cuda(rays_o).float().contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).float().contiguous().view(-1, 3)
        dev = rays_o.device
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs = torch.zeros(M, 3, device=dev)
        dirs = torch.zeros(M, 3, device=dev)
        deltas = torch.zeros(M, 2, device=dev)
        ops.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                       float(bound), float(dt_gamma), density_grid,
                       float(mean_density), near, far, xyzs, dirs, deltas,
                       int(perturb))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas,
                   weights_sum, depth, image, num_semantics_classes=None):
    """reference raymarching.py:455-504; in place on rays_t, weights_sum,
    depth, image.  (`num_semantics_classes` is accepted and unused, as in the
    reference's kernel.)"""
    with torch.no_grad():
        ops.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs,
                           None, deltas, weights_sum, depth, image, None)
    return tuple()


def composite_rays_semantics(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs,
                             local_semantics, deltas, weights_sum, depth,
                             image, semantics):
    """reference raymarching.py:507-558; also in place on semantics [N,C]."""
    with torch.no_grad():
        ops.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs,
                           local_semantics, deltas, weights_sum, depth, image,
                           semantics)
    return tuple()


def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old,
                 alive_counter):
    """reference raymarching.py:561-595; in place on rays_alive, rays_t and
    alive_counter (which is advanced by the number of survivors)."""
    with torch.no_grad():
        ops.compact_rays(n_alive, rays_alive, rays_alive_old, rays_t,
                         rays_t_old, alive_counter)
    return tuple()

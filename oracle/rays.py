"""Oracle (test infrastructure): ray generation and AABB clipping on the CPU.

* ``pixel_rays`` / ``pixel_rays_train`` follow reference
  ``nr4seg/dataset/ngp_utils.py:28-69`` and
  ``nr4seg/lightning/joint_train_lightning_net.py:108-157``.
* ``nerf_matrix_to_ngp`` follows ``nr4seg/dataset/ngp_utils.py:7-17``.
* ``near_far_from_aabb`` follows the CUDA kernel
  ``nr4seg/nerf/raymarching/src/raymarching.cu:62-115`` (wrapper default
  ``min_near=0.2``, ``nr4seg/nerf/raymarching/raymarching.py:16``).
"""
from __future__ import annotations

import numpy as np
import torch

FLT_MAX = float(np.finfo(np.float32).max)


def nerf_matrix_to_ngp(pose: np.ndarray) -> np.ndarray:
    p = np.asarray(pose)
    out = np.eye(4, dtype=np.float32)
    for r, src in enumerate((1, 2, 0)):
        out[r, 0] = p[src, 0]
        out[r, 1] = -p[src, 1]
        out[r, 2] = -p[src, 2]
        out[r, 3] = p[src, 3]
    return out


def _pixel_centres(H: int, W: int):
    # row-major pixel order; x = column + .5, y = row + .5
    cols = torch.linspace(0, W - 1, W)
    rows = torch.linspace(0, H - 1, H)
    px = cols.view(1, W).expand(H, W).reshape(-1) + 0.5
    py = rows.view(H, 1).expand(H, W).reshape(-1) + 0.5
    return px, py


def _rays_from_pixels(poses, intrinsics, px, py):
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    ones = torch.ones_like(px)
    dirs = torch.stack(((px - cx) / fx * ones, (py - cy) / fy * ones, ones),
                       dim=-1)
    norms = torch.norm(dirs, dim=-1, keepdim=True)
    dirs = dirs / norms
    rays_d = dirs @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o, rays_d, norms


def pixel_rays(poses: torch.Tensor, intrinsics, H: int, W: int):
    """poses [B,4,4] -> rays_o, rays_d [B,H*W,3], direction_norms [B,H*W,1]."""
    B = poses.shape[0]
    px, py = _pixel_centres(H, W)
    px = px.view(1, -1).expand(B, -1)
    py = py.view(1, -1).expand(B, -1)
    return _rays_from_pixels(poses, intrinsics, px, py)


def pixel_rays_train(poses: torch.Tensor, intrinsics, H: int, W: int,
                     inds: torch.Tensor):
    """Training variant: ``inds`` [N] int64 (the reference draws them with
    torch.randint, duplicates allowed) shared by every pose in the batch."""
    B = poses.shape[0]
    px, py = _pixel_centres(H, W)
    px = px[inds].view(1, -1).expand(B, -1)
    py = py[inds].view(1, -1).expand(B, -1)
    o, d, n = _rays_from_pixels(poses, intrinsics, px, py)
    return o, d, n, inds.view(1, -1).expand(B, -1)


def near_far_from_aabb(rays_o: torch.Tensor, rays_d: torch.Tensor,
                       aabb: torch.Tensor, min_near: float = 0.2):
    """Per-ray slab test, x then y then z, scalar fp32 semantics."""
    o = rays_o.reshape(-1, 3).numpy().astype(np.float32)
    d = rays_d.reshape(-1, 3).numpy().astype(np.float32)
    bb = aabb.numpy().astype(np.float32)
    N = o.shape[0]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        rd = (np.float32(1.0) / d).astype(np.float32)
        lo = ((bb[None, :3] - o) * rd).astype(np.float32)
        hi = ((bb[None, 3:] - o) * rd).astype(np.float32)
    # swap where near > far (a NaN compares false, as in C)
    sw = lo > hi
    lo2 = np.where(sw, hi, lo)
    hi2 = np.where(sw, lo, hi)
    near = lo2[:, 0].copy()
    far = hi2[:, 0].copy()
    miss = np.zeros(N, dtype=bool)
    for ax in (1, 2):
        n_a, f_a = lo2[:, ax], hi2[:, ax]
        miss_here = (~miss) & ((near > f_a) | (n_a > far))
        miss |= miss_here
        upd = ~miss
        near = np.where(upd & (n_a > near), n_a, near)
        far = np.where(upd & (f_a < far), f_a, far)
    near = np.where(near < np.float32(min_near), np.float32(min_near), near)
    near = np.where(miss, np.float32(FLT_MAX), near).astype(np.float32)
    far = np.where(miss, np.float32(FLT_MAX), far).astype(np.float32)
    return torch.from_numpy(near), torch.from_numpy(far)

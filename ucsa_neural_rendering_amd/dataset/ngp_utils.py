"""Mirror of reference ``nr4seg/dataset/ngp_utils.py``: ``get_rays`` and
``nerf_matrix_to_ngp``."""
import numpy as np
import torch

from .. import ops


def nerf_matrix_to_ngp(pose):
    """reference :7-17 -- row permutation (y, z, x) with sign flips."""
    p = np.asarray(pose)
    return np.array(
        [[p[1, 0], -p[1, 1], -p[1, 2], p[1, 3]],
         [p[2, 0], -p[2, 1], -p[2, 2], p[2, 3]],
         [p[0, 0], -p[0, 1], -p[0, 2], p[0, 3]],
         [0, 0, 0, 1]], dtype=np.float32)


@torch.no_grad()
def get_rays(poses, intrinsics, H, W, error_map=None):
    """reference :28-69.  poses [B,4,4] on the GPU -> dict(rays_o, rays_d
    [B,H*W,3], direction_norms [B,H*W,1]); the HIP kernel ``ucsa_get_rays``."""
    o, d, n = ops.get_rays(poses, intrinsics, int(H), int(W))
    return {"rays_o": o, "rays_d": d, "direction_norms": n}

"""Tensor-level wrappers over the C ABI (``include/ucsa_hip.h``).

PyTorch is used for device memory and streams only: every function checks its
arguments, allocates outputs with ``torch.empty`` and enqueues the HIP kernels
on torch's current stream.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import Grid, check, fvec, lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_current_stream() -> int:
    if _raw_stream is not None:
        return int(_raw_stream(torch.cuda.current_device()))
    return int(torch.cuda.current_stream().cuda_stream)


def _stream():
    # torch's current stream on the current device; the raw getter is ~20x
    # cheaper than torch.cuda.current_stream() (a training step makes ~17 calls)
    return C.c_void_p(_raw_current_stream())


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.UcsaError(
            f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.device.index != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's current stream
        raise _lib.UcsaError(
            f"{name} is on {t.device} but the current device is cuda:"
            f"{torch.cuda.current_device()}: call torch.cuda.set_device first")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def get_rays(poses: torch.Tensor, intrinsics, H: int, W: int,
             inds: Optional[torch.Tensor] = None):
    """-> rays_o, rays_d [B,n,3], direction_norms [B,n,1]."""
    poses = _f32(poses, "poses")
    B = poses.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    if inds is not None:
        inds = inds.reshape(-1).to(torch.int64).contiguous()
        n = inds.numel()
    else:
        n = H * W
    o = torch.empty(B, n, 3, device=poses.device)
    d = torch.empty(B, n, 3, device=poses.device)
    nr = torch.empty(B, n, 1, device=poses.device)
    check(lib().ucsa_get_rays(_ptr(poses), B, fx, fy, cx, cy, H, W,
                              _ptr(inds), n, _ptr(o), _ptr(d), _ptr(nr),
                              _stream()), "ucsa_get_rays")
    return o, d, nr


def tile_order(inds: torch.Tensor, W: int, tile: int = 16,
               H: Optional[int] = None) -> torch.Tensor:
    """The same pixel indices (duplicates kept), ordered tile by tile
    (``tile`` x ``tile`` pixels, row-major inside a tile).  A training batch
    is a random subset of one image; with neighbouring pixels next to each
    other the 32 rays of a hash-grid-backward workgroup share their coarse
    cells (k_hashgrid_bwd<true>: 0.86 -> 0.37 ms per pass, the step 7.9 ->
    6.9 ms).  The loss is a mean over the batch, so the order is free.

    GPU batches of <= 8192 indices: one launch (``ucsa_tile_order``, LDS
    bitonic sort of the 32-bit tile keys); otherwise torch ops (the key is a
    bijection of the pixel index, so both give the same result)."""
    flat = inds.reshape(-1)
    if flat.is_cuda and flat.dtype == torch.int64 and 0 < flat.numel() <= 8192:
        flat = flat.contiguous()
        out = torch.empty_like(flat)
        # H only bounds the key range; any image containing the indices will do
        rows = int(H) if H is not None else min((0x7FFFFFFE // int(W)) // 2, 1 << 20)
        check(lib().ucsa_tile_order(_ptr(flat), flat.numel(), rows, int(W),
                                    int(tile), _ptr(out), _stream()),
              "ucsa_tile_order")
        return out.reshape(inds.shape)
    y, x = flat // W, flat % W
    key = ((y // tile) * ((W + tile - 1) // tile) + x // tile) * (tile * tile) \
        + (y % tile) * tile + x % tile
    return flat[torch.argsort(key)].reshape(inds.shape)


def near_far_from_aabb(rays_o, rays_d, aabb, min_near: float = 0.2):
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    N = rays_o.shape[0]
    aabb_h = fvec(aabb.detach().cpu().tolist() if torch.is_tensor(aabb) else aabb)
    nears = torch.empty(N, device=rays_o.device)
    fars = torch.empty(N, device=rays_o.device)
    check(lib().ucsa_near_far_from_aabb(_ptr(rays_o), _ptr(rays_d), aabb_h, N,
                                        float(min_near), _ptr(nears),
                                        _ptr(fars), _stream()),
          "ucsa_near_far_from_aabb")
    return nears, fars


def sample_coarse(nears, fars, T: int, t_rand=None):
    nears = _f32(nears, "nears").view(-1)
    fars = _f32(fars, "fars").view(-1)
    N = nears.shape[0]
    if t_rand is not None:
        t_rand = _f32(t_rand, "t_rand")
        assert t_rand.shape == (N, T)
    z = torch.empty(N, T, device=nears.device)
    check(lib().ucsa_sample_coarse(_ptr(nears), _ptr(fars), _ptr(t_rand), N, T,
                                   _ptr(z), _stream()), "ucsa_sample_coarse")
    return z


def hashgrid_encode_rays(grid: Grid, table, rays_o, rays_d, z, aabb,
                         image_width: int = 0, half_features: bool = False):
    """-> feat [L, N*T, 2] (level-major).  image_width > 0: the rays are the
    pixels of full image rows (tile-ordered gather, same result).  An fp16
    table (table_to_half) or half_features=True gives fp16 features (for
    sigma_mlp_fwd_f16)."""
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    z = _f32(z, "z")
    N, T = z.shape
    if table.dtype == torch.float16:   # fp16 table (table_to_half) -> fp16 features
        feat = torch.empty(grid.n_levels, N * T, 2, dtype=torch.float16,
                           device=z.device)
        check(lib().ucsa_hashgrid_encode_rays_h16(
            C.byref(grid), _ptr(table), _ptr(rays_o), _ptr(rays_d), _ptr(z),
            fvec(aabb), N, T, int(image_width), _ptr(feat), _stream()),
            "ucsa_hashgrid_encode_rays_h16")
        return feat
    if half_features:
        feat = torch.empty(grid.n_levels, N * T, 2, dtype=torch.float16,
                           device=z.device)
        check(lib().ucsa_hashgrid_encode_rays_hf(
            C.byref(grid), _ptr(table), _ptr(rays_o), _ptr(rays_d), _ptr(z),
            fvec(aabb), N, T, int(image_width), _ptr(feat), _stream()),
            "ucsa_hashgrid_encode_rays_hf")
        return feat
    feat = torch.empty(grid.n_levels, N * T, 2, device=z.device)
    if image_width:
        check(lib().ucsa_hashgrid_encode_rays_image(
            C.byref(grid), _ptr(table), _ptr(rays_o), _ptr(rays_d), _ptr(z),
            fvec(aabb), N, T, int(image_width), _ptr(feat), _stream()),
            "ucsa_hashgrid_encode_rays_image")
        return feat
    check(lib().ucsa_hashgrid_encode_rays(C.byref(grid), _ptr(table),
                                          _ptr(rays_o), _ptr(rays_d), _ptr(z),
                                          fvec(aabb), N, T, _ptr(feat),
                                          _stream()),
          "ucsa_hashgrid_encode_rays")
    return feat


def tile_depth_order(z, image_width: int):
    """z [N,T] of image-ordered rays (whole rows, N % image_width == 0) ->
    (z_sorted [N*T] f32, pix [N*T] u8, slot [N*T] int32): every 8x8 pixel
    tile's samples in depth order, tiles back to back; ``slot`` = the ray-major
    index r * T + s of each (ucsa_tile_depth_order)."""
    z = _f32(z, "z")
    N, T = z.shape
    z_sorted = torch.empty(N * T, device=z.device)
    pix = torch.empty(N * T, dtype=torch.uint8, device=z.device)
    slot = torch.empty(N * T, dtype=torch.int32, device=z.device)
    check(lib().ucsa_tile_depth_order(_ptr(z), N, T, int(image_width), _ptr(z_sorted),
                                      _ptr(pix), _ptr(slot), _stream()),
          "ucsa_tile_depth_order")
    return z_sorted, pix, slot


def tile_index_order(z, image_width: int):
    """The arrays of ``tile_depth_order`` for COARSE samples without a sort: every
    8x8 tile's samples ordered by (sample index, pixel) (ucsa_tile_index_order)."""
    z = _f32(z, "z")
    N, T = z.shape
    z_sorted = torch.empty(N * T, device=z.device)
    pix = torch.empty(N * T, dtype=torch.uint8, device=z.device)
    slot = torch.empty(N * T, dtype=torch.int32, device=z.device)
    check(lib().ucsa_tile_index_order(_ptr(z), N, T, int(image_width), _ptr(z_sorted),
                                      _ptr(pix), _ptr(slot), _stream()),
          "ucsa_tile_index_order")
    return z_sorted, pix, slot


def hashgrid_encode_sorted(grid: Grid, table, rays_o, rays_d, z_sorted, pix, aabb,
                           T: int, image_width: int, half_features: bool = False):
    """-> feat [L, N*T, 2] in the order of ``tile_depth_order`` (fp32 table)."""
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    N = rays_o.shape[0]
    z_sorted = _f32(z_sorted, "z_sorted")
    if z_sorted.numel() != N * T or pix.numel() != N * T or pix.dtype != torch.uint8:
        raise UcsaError("hashgrid_encode_sorted: z_sorted / pix must hold N*T entries (pix uint8)")
    feat = torch.empty(grid.n_levels, N * T, 2, device=z_sorted.device,
                       dtype=torch.float16 if half_features else torch.float32)
    fn = (lib().ucsa_hashgrid_encode_sorted_hf if half_features
          else lib().ucsa_hashgrid_encode_sorted)
    check(fn(C.byref(grid), _ptr(table), _ptr(rays_o), _ptr(rays_d), _ptr(z_sorted),
             _ptr(pix), fvec(aabb), N, int(T), int(image_width), _ptr(feat), _stream()),
          "ucsa_hashgrid_encode_sorted")
    return feat


def sigma_mlp_fwd_scatter(mode: int, feat, packed_sigma, slot):
    """The sigma MLP (mode 0 f32-input MFMA, 1 f16 nets on fp16 features,
    2 bf16x3, 3 f16x2) on a depth-ordered feature array; h [M,16] / sigma [M]
    land at the ray-major rows ``slot``."""
    L, M, _ = feat.shape
    if slot.numel() != M or slot.dtype != torch.int32:
        raise UcsaError("sigma_mlp_fwd_scatter: slot must be int32 [M]")
    h = torch.empty(M, 16, device=feat.device)
    sigma = torch.empty(M, device=feat.device)
    check(lib().ucsa_sigma_mlp_fwd_scatter(int(mode), _ptr(feat), _ptr(packed_sigma), M, L,
                                           _ptr(slot), _ptr(h), _ptr(sigma), _stream()),
          "ucsa_sigma_mlp_fwd_scatter")
    return h, sigma


def density_sorted(mode: int, grid: Grid, table, rays_o, rays_d, z_sorted, pix, slot,
                   aabb, T: int, image_width: int, packed_sigma):
    """hashgrid_encode_sorted + sigma_mlp_fwd_scatter in one call, levels 0-7
    encoded inside the sigma MLP (ucsa_density_sorted; mode 2 bf16x3, 3 f16x2)
    -> h [N*T,16], sigma [N*T] at the ray-major rows."""
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    N = rays_o.shape[0]
    M = N * int(T)
    if z_sorted.numel() != M or pix.numel() != M or slot.numel() != M:
        raise _lib.UcsaError("density_sorted: z_sorted / pix / slot must hold N*T entries")
    feat_ws = torch.empty(grid.n_levels, M, 2, device=z_sorted.device)
    h = torch.empty(M, 16, device=z_sorted.device)
    sigma = torch.empty(M, device=z_sorted.device)
    check(lib().ucsa_density_sorted(
        int(mode), C.byref(grid), _ptr(table), _ptr(rays_o), _ptr(rays_d), _ptr(z_sorted),
        _ptr(pix), _ptr(slot), fvec(aabb), N, int(T), int(image_width), _ptr(packed_sigma),
        _ptr(feat_ws), _ptr(h), _ptr(sigma), _stream()), "ucsa_density_sorted")
    return h, sigma


def env_reload():
    """Make the library re-read its UCSA_* tuning switches (it snapshots them once
    per process: INTEGRATION.md "Environment variables").  Lab tools and tests
    that flip a switch inside one process call this after changing os.environ."""
    lib().ucsa_env_reload()


def hashgrid_encode_points(grid: Grid, table, x):
    x = _f32(x, "x").view(-1, 3)
    M = x.shape[0]
    feat = torch.empty(grid.n_levels, M, 2, device=x.device)
    check(lib().ucsa_hashgrid_encode_points(C.byref(grid), _ptr(table), _ptr(x),
                                            M, _ptr(feat), _stream()),
          "ucsa_hashgrid_encode_points")
    return feat


def mlp_pack(kind: int, params: torch.Tensor, n_classes: int = 0,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    params = _f32(params.detach(), "params")
    if out is None:
        out = torch.empty_like(params)
    check(lib().ucsa_mlp_pack(kind, _ptr(params), _ptr(out), n_classes,
                              _stream()), "ucsa_mlp_pack")
    return out


def sigma_mlp_fwd(feat, packed_sigma) -> Tuple[torch.Tensor, torch.Tensor]:
    """feat [L,M,2] -> h [M,16], sigma [M]."""
    L, M, _ = feat.shape
    h = torch.empty(M, 16, device=feat.device)
    sigma = torch.empty(M, device=feat.device)
    check(lib().ucsa_sigma_mlp_fwd(_ptr(feat), _ptr(packed_sigma), M, L,
                                   _ptr(h), _ptr(sigma), _stream()),
          "ucsa_sigma_mlp_fwd")
    return h, sigma


def resample(z, sigma, u, density_scale: float = 1.0):
    z = _f32(z, "z")
    sigma = _f32(sigma, "sigma")
    u = _f32(u, "u")
    N, T = z.shape
    t = u.shape[1]
    new_z = torch.empty(N, t, device=z.device)
    check(lib().ucsa_resample(_ptr(z), _ptr(sigma), _ptr(u), N, T, t,
                              float(density_scale), _ptr(new_z), _stream()),
          "ucsa_resample")
    return new_z


def composite_fwd(rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                  packed_color, packed_sem, n_classes: int,
                  density_scale: float = 1.0, want_aux: bool = False,
                  half: bool = False):
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    norms = _f32(norms, "norms").view(-1)
    N, T = z_c.shape
    t = 0 if z_f is None else z_f.shape[1]
    dev = z_c.device
    image = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    sem = torch.empty(N, n_classes, device=dev)
    src = torch.empty(N, T + t, dtype=torch.int32, device=dev) if want_aux else None
    w = torch.empty(N, T + t, device=dev) if want_aux else None
    fn = lib().ucsa_composite_fwd_f16 if half else lib().ucsa_composite_fwd
    check(fn(_ptr(rays_d), _ptr(norms), _ptr(z_c), _ptr(sigma_c), _ptr(h_c),
             _ptr(z_f), _ptr(sigma_f), _ptr(h_f), _ptr(packed_color),
             _ptr(packed_sem), N, T, t, n_classes, float(density_scale),
             _ptr(image), _ptr(depth), _ptr(sem), _ptr(src), _ptr(w), _stream()),
          "ucsa_composite_fwd")
    if want_aux:
        return image, depth, sem, src, w
    return image, depth, sem


def composite_train_fwd_x3(rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                           packed_color_x3, packed_sem_x3, n_classes: int,
                           density_scale: float = 1.0, h2: bool = False):
    """Training forward of the colour / semantics stage on the split pair with
    the bf16x3 nets -> (image, depth, sem, src, w) like
    composite_fwd(want_aux=True)."""
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    norms = _f32(norms, "norms").view(-1)
    N, T = z_c.shape
    t = 0 if z_f is None else z_f.shape[1]
    dev = z_c.device
    image = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    sem = torch.empty(N, n_classes, device=dev)
    src = torch.empty(N, T + t, dtype=torch.int32, device=dev)
    w = torch.empty(N, T + t, device=dev)
    ws = _scratch_named("composite_infer",
                        int(lib().ucsa_composite_infer_workspace_bytes(N, T, t)),
                        dev)
    fn = lib().ucsa_composite_train_fwd_h2 if h2 else lib().ucsa_composite_train_fwd_x3
    check(fn(
        _ptr(rays_d), _ptr(norms), _ptr(z_c), _ptr(sigma_c), _ptr(h_c), _ptr(z_f),
        _ptr(sigma_f), _ptr(h_f), _ptr(packed_color_x3), _ptr(packed_sem_x3), N, T,
        t, n_classes, float(density_scale), _ptr(image), _ptr(depth), _ptr(sem),
        _ptr(src), _ptr(w), _ptr(ws), _stream()), "ucsa_composite_train_fwd_x3")
    return image, depth, sem, src, w


def composite_infer(rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                    packed_color, packed_sem, n_classes: int,
                    density_scale: float = 1.0, half: bool = False,
                    x3: bool = False, h2: bool = False):
    """Inference composite as the dense kernel pair (ucsa_composite_infer):
    same outputs as composite_fwd, bit for bit.  half=True: packed weights
    from mlp_pack_f16.  x3=True: bf16x3 nets (fp32-grade, weights from
    mlp_pack_x3; not bit-identical to the f32-input MFMA); h2=True: f16x2
    nets (mlp_pack_h2)."""
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    norms = _f32(norms, "norms").view(-1)
    N, T = z_c.shape
    t = 0 if z_f is None else z_f.shape[1]
    dev = z_c.device
    image = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    sem = torch.empty(N, n_classes, device=dev)
    ws = _scratch_named("composite_infer",
                        int(lib().ucsa_composite_infer_workspace_bytes(N, T, t)),
                        dev)
    fn = (lib().ucsa_composite_infer_h2 if h2 else lib().ucsa_composite_infer_x3 if x3 else
          lib().ucsa_composite_infer_f16 if half else lib().ucsa_composite_infer)
    check(fn(_ptr(rays_d), _ptr(norms), _ptr(z_c), _ptr(sigma_c), _ptr(h_c),
             _ptr(z_f), _ptr(sigma_f), _ptr(h_f), _ptr(packed_color),
             _ptr(packed_sem), N, T, t, n_classes, float(density_scale),
             _ptr(image), _ptr(depth), _ptr(sem), _ptr(ws), _stream()),
          "ucsa_composite_infer")
    return image, depth, sem


def point_shade(dirs, geo_feat, mask, packed_color, packed_sem, n_classes: int,
                want_rgb: bool = True, want_probs: bool = True):
    geo_feat = _f32(geo_feat, "geo_feat").view(-1, 15)
    M = geo_feat.shape[0]
    dev = geo_feat.device
    dirs = None if dirs is None else _f32(dirs, "d").view(M, 3)
    m8 = None if mask is None else mask.reshape(M).to(torch.uint8).contiguous()
    rgb = torch.empty(M, 3, device=dev) if want_rgb else None
    probs = torch.empty(M, n_classes, device=dev) if want_probs else None
    check(lib().ucsa_point_shade(_ptr(dirs), _ptr(geo_feat), _ptr(m8),
                                 _ptr(packed_color), _ptr(packed_sem), M,
                                 n_classes, _ptr(rgb), _ptr(probs), _stream()),
          "ucsa_point_shade")
    return rgb, probs


def point_shade_h(dirs, h, packed_color, packed_sem, n_classes: int,
                  rgb=None, probs=None):
    """colour + semantics of M points straight from the sigma-MLP rows
    h [M,16]; optional preallocated outputs."""
    M = h.shape[0]
    dev = h.device
    dirs = _f32(dirs, "d").view(-1, 3)
    if rgb is None:
        rgb = torch.empty(M, 3, device=dev)
    if probs is None:
        probs = torch.empty(M, n_classes, device=dev)
    check(lib().ucsa_point_shade_h(_ptr(dirs), _ptr(h), None,
                                   _ptr(packed_color), _ptr(packed_sem), M,
                                   n_classes, _ptr(rgb), _ptr(probs),
                                   _stream()), "ucsa_point_shade_h")
    return rgb, probs


def render_workspace_bytes(N: int, T: int, t: int, n_levels: int) -> int:
    return int(lib().ucsa_render_workspace_bytes(N, T, t, n_levels))


def render_fwd(grid: Grid, table, packed_sigma, packed_color, packed_sem,
               rays_o, rays_d, norms, aabb, min_near: float, t_rand, u, T: int,
               t: int, n_classes: int, density_scale: float, image, depth,
               semantics, ws: torch.Tensor, image_width: int = 0):
    """All tensors already validated/contiguous; outputs written in place.
    image_width > 0: the rays are the pixels of full image rows."""
    N = rays_o.shape[0]
    check(lib().ucsa_render_fwd(C.byref(grid), _ptr(table), _ptr(packed_sigma),
                                _ptr(packed_color), _ptr(packed_sem),
                                _ptr(rays_o), _ptr(rays_d), _ptr(norms),
                                fvec(aabb), float(min_near), _ptr(t_rand),
                                _ptr(u), N, T, t, n_classes,
                                float(density_scale), int(image_width),
                                _ptr(image), _ptr(depth), _ptr(semantics),
                                _ptr(ws), _stream()),
          "ucsa_render_fwd")


# ======================= fp16-MFMA inference option =========================
def mlp_pack_f16(kind: int, params: torch.Tensor, n_classes: int = 0,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    params = _f32(params.detach(), "params")
    n = int(lib().ucsa_mlp_pack_f16_halves(kind, n_classes))
    if out is None:
        out = torch.empty(n, dtype=torch.float16, device=params.device)
    check(lib().ucsa_mlp_pack_f16(kind, _ptr(params), _ptr(out), n_classes,
                                  _stream()), "ucsa_mlp_pack_f16")
    return out


def mlp_pack_x3(kind: int, params: torch.Tensor, n_classes: int = 0,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weights as three bf16 terms per value in MFMA fragment order
    (csrc/mfma_mlp_x3.h)."""
    params = _f32(params.detach(), "params")
    n = int(lib().ucsa_mlp_pack_x3_bytes(kind, n_classes))
    if out is None:
        out = torch.empty(n // 2, dtype=torch.bfloat16, device=params.device)
    check(lib().ucsa_mlp_pack_x3(kind, _ptr(params), _ptr(out), n_classes,
                                 _stream()), "ucsa_mlp_pack_x3")
    return out


def mlp_pack_h2(kind: int, params: torch.Tensor, n_classes: int = 0,
                out: Optional[torch.Tensor] = None,
                range_bits: Optional[torch.Tensor] = None,
                scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weights as two f16 terms per value (the second scaled by 2^11) in MFMA
    fragment order (csrc/mfma_mlp_h2.h, "f16x2").  ``range_bits`` (int32 [1] in
    pinned host or device memory, zeroed by the caller) accumulates the largest
    |value| converted to f16 as an fp32 bit pattern; ``scratch``: int32 [2] on
    the device, zeroed once (ucsa_mlp_pack_h2_checked)."""
    params = _f32(params.detach(), "params")
    n = int(lib().ucsa_mlp_pack_h2_bytes(kind, n_classes))
    if out is None:
        out = torch.empty(n // 2, dtype=torch.float16, device=params.device)
    if range_bits is not None:
        if scratch is None or not scratch.is_cuda or scratch.numel() < 2:
            raise UcsaError("mlp_pack_h2: range_bits needs a 2-word int32 device scratch")
        check(lib().ucsa_mlp_pack_h2_checked(kind, _ptr(params), _ptr(out), n_classes,
                                             _ptr(range_bits), _ptr(scratch), _stream()),
              "ucsa_mlp_pack_h2_checked")
        return out
    check(lib().ucsa_mlp_pack_h2(kind, _ptr(params), _ptr(out), n_classes,
                                 _stream()), "ucsa_mlp_pack_h2")
    return out


def sigma_mlp_fwd_h2(feat, packed_sigma_h2):
    L, M, _ = feat.shape
    h = torch.empty(M, 16, device=feat.device)
    sigma = torch.empty(M, device=feat.device)
    check(lib().ucsa_sigma_mlp_fwd_h2(_ptr(feat), _ptr(packed_sigma_h2), M, L,
                                      _ptr(h), _ptr(sigma), _stream()),
          "ucsa_sigma_mlp_fwd_h2")
    return h, sigma


def render_fwd_h2(grid: Grid, table, packed_sigma_h2, packed_color_h2,
                  packed_sem_h2, rays_o, rays_d, norms, aabb, min_near: float,
                  t_rand, u, T: int, t: int, n_classes: int,
                  density_scale: float, image, depth, semantics,
                  ws: torch.Tensor, image_width: int = 0):
    """run() with the three nets as f16x2 (fp32-grade on the f16 MFMA pipe,
    three passes per product)."""
    N = rays_o.shape[0]
    check(lib().ucsa_render_fwd_h2(
        C.byref(grid), _ptr(table), _ptr(packed_sigma_h2), _ptr(packed_color_h2),
        _ptr(packed_sem_h2), _ptr(rays_o), _ptr(rays_d), _ptr(norms), fvec(aabb),
        float(min_near), _ptr(t_rand), _ptr(u), N, T, t, n_classes,
        float(density_scale), int(image_width), _ptr(image), _ptr(depth),
        _ptr(semantics), _ptr(ws), _stream()), "ucsa_render_fwd_h2")


def sigma_mlp_fwd_f16(feat, packed_sigma_half):
    L, M, _ = feat.shape
    h = torch.empty(M, 16, device=feat.device)
    sigma = torch.empty(M, device=feat.device)
    if feat.dtype == torch.float16:   # features of the fp16-table encoder
        check(lib().ucsa_sigma_mlp_fwd_f16_h(_ptr(feat), _ptr(packed_sigma_half),
                                             M, L, _ptr(h), _ptr(sigma),
                                             _stream()), "ucsa_sigma_mlp_fwd_f16_h")
        return h, sigma
    check(lib().ucsa_sigma_mlp_fwd_f16(_ptr(feat), _ptr(packed_sigma_half), M,
                                       L, _ptr(h), _ptr(sigma), _stream()),
          "ucsa_sigma_mlp_fwd_f16")
    return h, sigma


def render_fwd_f16(grid: Grid, table, packed_sigma_h, packed_color_h,
                   packed_sem_h, rays_o, rays_d, norms, aabb, min_near: float,
                   t_rand, u, T: int, t: int, n_classes: int,
                   density_scale: float, image, depth, semantics,
                   ws: torch.Tensor, image_width: int = 0):
    N = rays_o.shape[0]
    check(lib().ucsa_render_fwd_f16(
        C.byref(grid), _ptr(table), _ptr(packed_sigma_h), _ptr(packed_color_h),
        _ptr(packed_sem_h), _ptr(rays_o), _ptr(rays_d), _ptr(norms), fvec(aabb),
        float(min_near), _ptr(t_rand), _ptr(u), N, T, t, n_classes,
        float(density_scale), int(image_width), _ptr(image), _ptr(depth),
        _ptr(semantics), _ptr(ws), _stream()), "ucsa_render_fwd_f16")


def table_to_half(table: torch.Tensor, out: Optional[torch.Tensor] = None):
    """fp32 hash table -> the fp16 copy the *_h16 entry points read."""
    table = _f32(table.detach(), "table")
    if out is None:
        out = torch.empty(table.numel(), dtype=torch.float16, device=table.device)
    check(lib().ucsa_cast_f32_to_f16(_ptr(table), _ptr(out), table.numel(),
                                     _stream()), "ucsa_cast_f32_to_f16")
    return out


def render_fwd_f16_h16(grid: Grid, table_half, packed_sigma_h, packed_color_h,
                       packed_sem_h, rays_o, rays_d, norms, aabb, min_near: float,
                       t_rand, u, T: int, t: int, n_classes: int,
                       density_scale: float, image, depth, semantics,
                       ws: torch.Tensor, image_width: int = 0):
    """render_fwd_f16 reading the fp16 table (table_to_half)."""
    N = rays_o.shape[0]
    assert table_half.dtype == torch.float16
    check(lib().ucsa_render_fwd_f16_h16(
        C.byref(grid), _ptr(table_half), _ptr(packed_sigma_h), _ptr(packed_color_h),
        _ptr(packed_sem_h), _ptr(rays_o), _ptr(rays_d), _ptr(norms), fvec(aabb),
        float(min_near), _ptr(t_rand), _ptr(u), N, T, t, n_classes,
        float(density_scale), int(image_width), _ptr(image), _ptr(depth),
        _ptr(semantics), _ptr(ws), _stream()), "ucsa_render_fwd_f16_h16")


def sigma_mlp_fwd_x3(feat, packed_sigma_x3):
    L, M, _ = feat.shape
    h = torch.empty(M, 16, device=feat.device)
    sigma = torch.empty(M, device=feat.device)
    check(lib().ucsa_sigma_mlp_fwd_x3(_ptr(feat), _ptr(packed_sigma_x3), M, L,
                                      _ptr(h), _ptr(sigma), _stream()),
          "ucsa_sigma_mlp_fwd_x3")
    return h, sigma


def render_fwd_x3(grid: Grid, table, packed_sigma_x3, packed_color_x3,
                  packed_sem_x3, rays_o, rays_d, norms, aabb, min_near: float,
                  t_rand, u, T: int, t: int, n_classes: int,
                  density_scale: float, image, depth, semantics,
                  ws: torch.Tensor, image_width: int = 0):
    """run() with the three nets as bf16x3 (fp32-grade on the bf16 MFMA pipe)."""
    N = rays_o.shape[0]
    check(lib().ucsa_render_fwd_x3(
        C.byref(grid), _ptr(table), _ptr(packed_sigma_x3), _ptr(packed_color_x3),
        _ptr(packed_sem_x3), _ptr(rays_o), _ptr(rays_d), _ptr(norms), fvec(aabb),
        float(min_near), _ptr(t_rand), _ptr(u), N, T, t, n_classes,
        float(density_scale), int(image_width), _ptr(image), _ptr(depth),
        _ptr(semantics), _ptr(ws), _stream()), "ucsa_render_fwd_x3")


RENDER_MODES = {"fp32": 0, "fp16": 1, "bf16x3": 2, "fp16_h16": 3, "f16x2": 4}


def render_view(mode: str, grid: Grid, table, packed_sigma, packed_color, packed_sem,
                rays_o, rays_d, norms, aabb, min_near: float, t_rand, u, T: int,
                t: int, n_classes: int, density_scale: float, image, depth,
                semantics, chunk: int, ws0: torch.Tensor, ws1, image_width: int = 0):
    """All N rays in `chunk`-ray pieces as ONE call (ucsa_render_view): the
    density half of chunk k+1 overlaps the shading half of chunk k on two
    internal streams; ``ws1`` None = the serial loop.  Same bits either way."""
    N = rays_o.shape[0]
    check(lib().ucsa_render_view(
        RENDER_MODES[mode], C.byref(grid), _ptr(table), _ptr(packed_sigma),
        _ptr(packed_color), _ptr(packed_sem), _ptr(rays_o), _ptr(rays_d), _ptr(norms),
        fvec(aabb), float(min_near), _ptr(t_rand), _ptr(u), N, T, t, n_classes,
        float(density_scale), int(image_width), int(chunk), _ptr(image), _ptr(depth),
        _ptr(semantics), _ptr(ws0), _ptr(ws1), _stream()), "ucsa_render_view")


# ============================ training (backward) ===========================
def mlp_pack_t(kind: int, params: torch.Tensor, n_classes: int = 0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    params = _f32(params.detach(), "params")
    n = int(lib().ucsa_mlp_pack_t_size(kind, n_classes))
    if out is None:
        out = torch.empty(n, device=params.device)
    check(lib().ucsa_mlp_pack_t(kind, _ptr(params), _ptr(out), n_classes,
                                _stream()), "ucsa_mlp_pack_t")
    return out


def reduce_partials(partial: torch.Tensor, grad: torch.Tensor,
                    accumulate: bool):
    n_parts, n_params = partial.shape
    check(lib().ucsa_reduce_partials(_ptr(partial), n_parts, n_params,
                                     1 if accumulate else 0, _ptr(grad),
                                     _stream()), "ucsa_reduce_partials")


def reduce_partials_multi(pairs, accumulate: bool = False):
    """[(partial [parts, n], grad [n]), ...] (<= 4 pairs) in one launch."""
    k = len(pairs)
    P = (C.c_void_p * k)(*[p.data_ptr() for p, _ in pairs])
    G = (C.c_void_p * k)(*[g.data_ptr() for _, g in pairs])
    NP = (C.c_uint32 * k)(*[p.shape[0] for p, _ in pairs])
    NN = (C.c_uint32 * k)(*[p.shape[1] for p, _ in pairs])
    check(lib().ucsa_reduce_partials_multi(k, P, NP, NN, G,
                                           1 if accumulate else 0, _stream()),
          "ucsa_reduce_partials_multi")


def sigma_mlp_bwd(feat, d_h, packed_sigma, packed_sigma_t, x2: bool = False,
                  round_hidden: bool = False):
    """-> d_feat [L,M,2], dW partials [parts, 3072].  x2: the bf16x2 kernel
    (packed weights from mlp_pack_x3 / mlp_pack_t_x3); round_hidden: the fp32
    kernel with its recomputed hidden layer rounded to fp16 (tcnn numerics)."""
    L, M, _ = feat.shape
    d_feat = torch.empty_like(feat)
    parts = int(lib().ucsa_sigma_mlp_bwd_parts(M))
    partial = torch.empty(parts, 3072, device=feat.device)
    fn, nm = ((lib().ucsa_sigma_mlp_bwd_x2, "ucsa_sigma_mlp_bwd_x2") if x2 else
              (lib().ucsa_sigma_mlp_bwd_h16, "ucsa_sigma_mlp_bwd_h16") if round_hidden else
              (lib().ucsa_sigma_mlp_bwd, "ucsa_sigma_mlp_bwd"))
    check(fn(_ptr(feat), _ptr(d_h), _ptr(packed_sigma), _ptr(packed_sigma_t), M, L,
             _ptr(d_feat), _ptr(partial), _stream()), nm)
    return d_feat, partial


_bwd_ws = {}


def hashgrid_bwd_rays(grid: Grid, rays_o, rays_d, z, aabb, d_feat, grad_table,
                      binned: bool = True, rec_scale: float = 0.0,
                      packed: bool = False):
    """Adds the table gradient.  binned=True: two-pass LDS-binned algorithm
    (workspace cached per device); False: direct float atomics.
    rec_scale > 0 (binned only): 8-byte bin records, values as half2 x
    rec_scale (the f16 training mode).  packed (binned only): 8-byte records
    with 26-bit values (ucsa_hashgrid_bwd_rays_p64)."""
    N, T = z.shape
    if tuple(d_feat.shape) != (grid.n_levels, N * T, 2) or not d_feat.is_contiguous():
        raise _lib.UcsaError(
            f"d_feat must be a contiguous [{grid.n_levels}, {N * T}, 2] tensor "
            f"(levels, N*T samples of z {tuple(z.shape)}, 2), got {tuple(d_feat.shape)}")
    if tuple(rays_o.shape) != (N, 3) or tuple(rays_d.shape) != (N, 3):
        raise _lib.UcsaError("rays_o / rays_d must be [N, 3] with N = z.shape[0]")
    ws = None
    if binned:
        need = int(lib().ucsa_hashgrid_bwd_workspace_bytes(N, T, grid.n_levels))
        key = (z.device, _raw_current_stream())   # one bin buffer per stream
        ws = _bwd_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=z.device)
            _bwd_ws[key] = ws
    if rec_scale > 0.0 and ws is not None:
        check(lib().ucsa_hashgrid_bwd_rays_h16(
            C.byref(grid), _ptr(rays_o), _ptr(rays_d), _ptr(z), fvec(aabb), N, T,
            _ptr(d_feat), _ptr(grad_table), _ptr(ws), float(rec_scale), _stream()),
            "ucsa_hashgrid_bwd_rays_h16")
        return
    if packed and ws is not None:
        check(lib().ucsa_hashgrid_bwd_rays_p64(
            C.byref(grid), _ptr(rays_o), _ptr(rays_d), _ptr(z), fvec(aabb), N, T,
            _ptr(d_feat), _ptr(grad_table), _ptr(ws), _stream()),
            "ucsa_hashgrid_bwd_rays_p64")
        return
    check(lib().ucsa_hashgrid_bwd_rays(C.byref(grid), _ptr(rays_o),
                                       _ptr(rays_d), _ptr(z), fvec(aabb), N, T,
                                       _ptr(d_feat), _ptr(grad_table),
                                       _ptr(ws), _stream()),
          "ucsa_hashgrid_bwd_rays")


def hashgrid_bwd_rays_det(grid: Grid, rays_o, rays_d, z, aabb, d_feat, fix=None):
    """Deterministic accumulation of the table gradient of one density pass
    into the int64 fixed-point buffer ``fix`` (created zeroed when None;
    ucsa_hashgrid_bwd_rays_det).  Finish with ``hashgrid_bwd_det_finish``."""
    N, T = z.shape
    if tuple(d_feat.shape) != (grid.n_levels, N * T, 2) or not d_feat.is_contiguous():
        raise _lib.UcsaError("d_feat must be a contiguous [L, N*T, 2] tensor")
    if fix is None:
        n = int(lib().ucsa_hashgrid_bwd_det_workspace_bytes(C.byref(grid))) // 8
        fix = torch.zeros(n, dtype=torch.int64, device=z.device)
    check(lib().ucsa_hashgrid_bwd_rays_det(
        C.byref(grid), _ptr(rays_o), _ptr(rays_d), _ptr(z), fvec(aabb), N, T,
        _ptr(d_feat), _ptr(fix), _stream()), "ucsa_hashgrid_bwd_rays_det")
    return fix


def hashgrid_bwd_det_finish(grid: Grid, fix, grad_table):
    """grad_table += fix * 2^-44 (NaN everywhere after a non-finite contribution)."""
    check(lib().ucsa_hashgrid_bwd_det_finish(C.byref(grid), _ptr(fix), _ptr(grad_table),
                                             _stream()), "ucsa_hashgrid_bwd_det_finish")


def hashgrid_bwd_rays_merged(grid: Grid, rays_o, rays_d, z_c, z_f, src, aabb,
                             d_feat_c, d_feat_f, grad_table, packed: bool = False,
                             rec_scale: float = 0.0):
    """Both density passes in one call, every ray's samples walked in sorted
    depth order (``src`` [N, Tc+Tf] int32 of the forward composite): adds the
    table gradient (ucsa_hashgrid_bwd_rays_merged; packed: its _p64 form with
    8-byte records of 26-bit values)."""
    N, Tc = z_c.shape
    Tf = z_f.shape[1]
    L = grid.n_levels
    if (tuple(d_feat_c.shape) != (L, N * Tc, 2) or tuple(d_feat_f.shape) != (L, N * Tf, 2)
            or not d_feat_c.is_contiguous() or not d_feat_f.is_contiguous()):
        raise _lib.UcsaError("d_feat_c / d_feat_f must be contiguous [L, N*T, 2] tensors")
    if tuple(src.shape) != (N, Tc + Tf) or src.dtype != torch.int32 or not src.is_contiguous():
        raise _lib.UcsaError(f"src must be a contiguous int32 [{N}, {Tc + Tf}] tensor")
    if tuple(rays_o.shape) != (N, 3) or tuple(rays_d.shape) != (N, 3):
        raise _lib.UcsaError("rays_o / rays_d must be [N, 3] with N = z_c.shape[0]")
    need = int(lib().ucsa_hashgrid_bwd_workspace_bytes(N, Tc + Tf, L))
    key = (z_c.device, _raw_current_stream())
    ws = _bwd_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=z_c.device)
        _bwd_ws[key] = ws
    if rec_scale > 0.0:   # half2 x rec_scale records (training modes fp16 / tcnn)
        check(lib().ucsa_hashgrid_bwd_rays_merged_h16(
            C.byref(grid), _ptr(rays_o), _ptr(rays_d), _ptr(z_c), _ptr(z_f), _ptr(src),
            fvec(aabb), N, Tc, Tf, _ptr(d_feat_c), _ptr(d_feat_f), _ptr(grad_table), _ptr(ws),
            float(rec_scale), _stream()), "ucsa_hashgrid_bwd_rays_merged_h16")
        return
    fn = lib().ucsa_hashgrid_bwd_rays_merged_p64 if packed else lib().ucsa_hashgrid_bwd_rays_merged
    check(fn(
        C.byref(grid), _ptr(rays_o), _ptr(rays_d), _ptr(z_c), _ptr(z_f), _ptr(src),
        fvec(aabb), N, Tc, Tf, _ptr(d_feat_c), _ptr(d_feat_f), _ptr(grad_table), _ptr(ws),
        _stream()), "ucsa_hashgrid_bwd_rays_merged" + ("_p64" if packed else ""))


def train_packs(sigma_x3, color_x3, sem_x3, sigma_t_x3=None, color_t_x3=None,
                sem_t_x3=None, sigma_h2=None, color_h2=None, sem_h2=None) -> "_lib.TrainPacks":
    """ucsa_train_packs over bf16x3 weight fragments (mlp_pack_x3 / mlp_pack_t_x3)
    and, optionally, f16x2 ones for the forward (mlp_pack_h2); the caller keeps
    the tensors alive."""
    return _lib.TrainPacks(*[_ptr(x) for x in (sigma_x3, color_x3, sem_x3, sigma_t_x3,
                                              color_t_x3, sem_t_x3, sigma_h2, color_h2,
                                              sem_h2)])


def render_fused_fwd(grid: Grid, table, packs, rays_o, rays_d, norms, aabb,
                     min_near: float, t_rand, u, T: int, t: int, n_classes: int,
                     density_scale: float = 1.0):
    """The training forward (rows a2-a9) as ONE C call (ucsa_render_fused_fwd,
    bf16x3 nets) -> image [N,3], depth [N], sem [N,C] and the dict of saved
    tensors (the members of ucsa_train_buffers) the fused backward reads."""
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    norms = _f32(norms, "norms").view(-1)
    N, dev, L = rays_o.shape[0], rays_o.device, grid.n_levels
    if t_rand is not None:
        t_rand = _f32(t_rand, "t_rand")
        assert t_rand.shape == (N, T)
    if t > 0:
        u = _f32(u, "u")
        assert u.shape == (N, t)
    e = lambda *shape, **kw: torch.empty(*shape, device=dev, **kw)  # noqa: E731
    # The saved tensors of a step are carved from ONE allocation at fixed, skewed
    # offsets (array i starts 256 * (2 i + 1) bytes past a 4 KiB boundary).  As
    # ten separate torch.empty calls their relative placement was whatever the
    # caching allocator's history made it, and the step time followed it: +0.15 ms
    # (3.40 -> 3.55 ms) after ONE extra device synchronisation earlier in the
    # process (round 5, tools/gpu notes in the notebook); the C side already
    # skews its own workspace for the same reason (render_train.hip).
    shapes = [("z_c", (N, T), 4), ("feat_c", (L, N * T, 2), 4), ("h_c", (N * T, 16), 4),
              ("sigma_c", (N, T), 4)]
    if t:
        shapes += [("z_f", (N, t), 4), ("feat_f", (L, N * t, 2), 4), ("h_f", (N * t, 16), 4),
                   ("sigma_f", (N, t), 4)]
    shapes += [("src", (N, T + t), 4), ("weights", (N, T + t), 4)]
    offs, end = [], 0
    for i, (_, shp, b) in enumerate(shapes):
        start = (end + 4095) // 4096 * 4096 + 256 * (2 * i + 1)
        n = b
        for d_ in shp:
            n *= d_
        offs.append(start)
        end = start + n
    slab = torch.empty(end + 4096, dtype=torch.uint8, device=dev)
    base = (-slab.data_ptr()) % 4096          # 4 KiB-aligned origin inside the slab
    sv = {k: None for k in ("z_f", "feat_f", "h_f", "sigma_f")}
    for (k, shp, b), off in zip(shapes, offs):
        n = b
        for d_ in shp:
            n *= d_
        raw = slab[base + off: base + off + n]
        sv[k] = raw.view(torch.int32 if k == "src" else torch.float32).view(*shp)
    bufs = _lib.TrainBuffers(*[_ptr(sv[k]) for k, _ in _lib.TrainBuffers._fields_])
    image, depth, sem = e(N, 3), e(N), e(N, n_classes)
    ws = _scratch_named("render_fused_fwd",
                        int(lib().ucsa_render_fused_fwd_workspace_bytes(N, T, t)), dev)
    aabb_h = fvec(aabb.detach().cpu().tolist() if torch.is_tensor(aabb) else aabb)
    check(lib().ucsa_render_fused_fwd(
        C.byref(grid), _ptr(table), C.byref(packs), _ptr(rays_o), _ptr(rays_d), _ptr(norms),
        aabb_h, float(min_near), _ptr(t_rand), _ptr(u) if t else None, N, T, t, n_classes,
        float(density_scale), C.byref(bufs), _ptr(image), _ptr(depth), _ptr(sem), _ptr(ws),
        _stream()), "ucsa_render_fused_fwd")
    return image, depth, sem, sv


def render_fused_bwd(grid: Grid, packs, rays_o, rays_d, norms, aabb, saved: dict,
                     d_image, d_depth, d_sem, n_classes: int, density_scale: float,
                     grad_table, grad_sigma, grad_color, grad_sem):
    """The training backward as ONE C call (ucsa_render_fused_bwd: bf16x2
    contractions, packed bin records).  ADDS to grad_table, overwrites the three
    net gradients."""
    N, T = saved["z_c"].shape
    t = 0 if saved["z_f"] is None else saved["z_f"].shape[1]
    dev = saved["z_c"].device
    d_image = _f32(d_image, "d_image").view(N, 3)
    d_depth = _f32(d_depth, "d_depth").view(N)
    d_sem = _f32(d_sem, "d_sem").view(N, n_classes)
    nsem = 1024 + 1024 * ((n_classes + 15) // 16)
    assert grad_sigma.numel() == 3072 and grad_color.numel() == 7168 and grad_sem.numel() == nsem
    bufs = _lib.TrainBuffers(*[_ptr(saved[k]) for k, _ in _lib.TrainBuffers._fields_])
    ws = _scratch_named("render_fused_bwd", int(lib().ucsa_render_fused_bwd_workspace_bytes(
        N, T, t, n_classes, grid.n_levels)), dev)
    aabb_h = fvec(aabb.detach().cpu().tolist() if torch.is_tensor(aabb) else aabb)
    check(lib().ucsa_render_fused_bwd(
        C.byref(grid), C.byref(packs), _ptr(rays_o), _ptr(rays_d), _ptr(norms), aabb_h,
        C.byref(bufs), _ptr(d_image), _ptr(d_depth), _ptr(d_sem), N, T, t, n_classes,
        float(density_scale), _ptr(grad_table), _ptr(grad_sigma), _ptr(grad_color),
        _ptr(grad_sem), _ptr(ws), _stream()), "ucsa_render_fused_bwd")


def hashgrid_bwd_points(grid: Grid, x, d_feat, grad_table, binned: bool = True):
    """Backward of hashgrid_encode_points: adds into grad_table."""
    x = _f32(x, "x").view(-1, 3)
    M = x.shape[0]
    if tuple(d_feat.shape) != (grid.n_levels, M, 2) or not d_feat.is_contiguous():
        raise _lib.UcsaError(
            f"d_feat must be a contiguous [{grid.n_levels}, {M}, 2] tensor, got "
            f"{tuple(d_feat.shape)}")
    ws = None
    if binned:
        need = int(lib().ucsa_hashgrid_bwd_workspace_bytes(M, 1, grid.n_levels))
        key = (x.device, _raw_current_stream())
        ws = _bwd_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            _bwd_ws[key] = ws
    check(lib().ucsa_hashgrid_bwd_points(C.byref(grid), _ptr(x), M,
                                         _ptr(d_feat), _ptr(grad_table),
                                         _ptr(ws), _stream()),
          "ucsa_hashgrid_bwd_points")


def mlp_pack_t_f16(kind: int, params: torch.Tensor, n_classes: int = 0,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    params = _f32(params.detach(), "params")
    n = int(lib().ucsa_mlp_pack_t_f16_halves(kind, n_classes))
    if out is None:
        out = torch.empty(n, dtype=torch.float16, device=params.device)
    check(lib().ucsa_mlp_pack_t_f16(kind, _ptr(params), _ptr(out), n_classes,
                                    _stream()), "ucsa_mlp_pack_t_f16")
    return out


def mlp_pack_t_x3(kind: int, params: torch.Tensor, n_classes: int = 0,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Transposed weight fragments (dX = W^T dY) as three bf16 terms."""
    params = _f32(params.detach(), "params")
    n = int(lib().ucsa_mlp_pack_t_x3_bytes(kind, n_classes))
    if out is None:
        out = torch.empty(n // 2, dtype=torch.bfloat16, device=params.device)
    check(lib().ucsa_mlp_pack_t_x3(kind, _ptr(params), _ptr(out), n_classes,
                                   _stream()), "ucsa_mlp_pack_t_x3")
    return out


def shade_bwd_split() -> bool:
    """The colour / semantics backward runs as the per-net kernel pair
    (default; UCSA_SHADE_BWD_SPLIT=0 keeps the single kernel)."""
    return os.environ.get("UCSA_SHADE_BWD_SPLIT", "1")[:1] != "0"


def composite_bwd(rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f, src,
                  weights, packed_color, packed_sem, packed_color_t,
                  packed_sem_t, d_image, d_depth, d_sem, n_classes: int,
                  density_scale: float = 1.0, half: bool = False,
                  f16_scale: float = 1024.0, x2: bool = False):
    """-> d_h_c [N*T,16], d_h_f [N*t,16] | None, partial_color, partial_sem.
    half=True: packed weights from mlp_pack_f16 / mlp_pack_t_f16; x2=True:
    from mlp_pack_x3 / mlp_pack_t_x3 (bf16x2 contractions)."""
    N, T = z_c.shape
    t = 0 if z_f is None else z_f.shape[1]
    dev = z_c.device
    d_image = _f32(d_image, "d_image").view(N, 3)
    d_depth = _f32(d_depth, "d_depth").view(N)
    d_sem = _f32(d_sem, "d_sem").view(N, n_classes)
    G = torch.empty(N, T + t, device=dev)
    d_h_c = torch.empty(N * T, 16, device=dev)
    d_h_f = torch.empty(N * t, 16, device=dev) if t else None
    parts = int((lib().ucsa_composite_bwd_parts_f16 if half
                 else lib().ucsa_composite_bwd_parts)(N))
    nrb = (n_classes + 15) // 16
    pc = torch.empty(parts, 7168, device=dev)
    ps = torch.empty(parts, 1024 + 1024 * nrb, device=dev)
    head = (_ptr(rays_d), _ptr(norms), _ptr(z_c), _ptr(sigma_c), _ptr(h_c),
            _ptr(z_f), _ptr(sigma_f), _ptr(h_f), _ptr(src), _ptr(weights),
            _ptr(packed_color), _ptr(packed_sem), _ptr(packed_color_t),
            _ptr(packed_sem_t), _ptr(d_image), _ptr(d_depth), _ptr(d_sem), N, T,
            t, n_classes, float(density_scale))
    tail = (_ptr(G), _ptr(d_h_c), _ptr(d_h_f), _ptr(pc), _ptr(ps), _stream())
    if x2:
        check(lib().ucsa_composite_bwd_x2(*head, *tail), "ucsa_composite_bwd_x2")
    elif half:
        check(lib().ucsa_composite_bwd_f16(*head, float(f16_scale), *tail),
              "ucsa_composite_bwd_f16")
    else:
        check(lib().ucsa_composite_bwd(*head, *tail), "ucsa_composite_bwd")
    return d_h_c, d_h_f, pc, ps


def adam_step(p, g, m, v, step: int, lr: float, beta1: float, beta2: float,
              eps: float, weight_decay: float, inv_grad_scale: float = 1.0):
    for x in (p, g, m, v):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    check(lib().ucsa_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(),
                               int(step), float(lr), float(beta1), float(beta2),
                               float(eps), float(weight_decay),
                               float(inv_grad_scale), _stream()),
          "ucsa_adam_step")


def adam_step_scaled(p, g, m, v, step: int, lr: float, beta1: float,
                     beta2: float, eps: float, weight_decay: float,
                     grad_scale: torch.Tensor, found_inf: torch.Tensor,
                     skipped: torch.Tensor):
    """Adam step under GradScaler with the scale / found-inf flag on the
    device (no read-back); `skipped` int32[1] counts the skipped steps."""
    for x in (p, g, m, v):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    assert grad_scale.dtype == torch.float32 and found_inf.dtype == torch.float32
    assert skipped.dtype == torch.int32
    check(lib().ucsa_adam_step_scaled(
        _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(step), float(lr),
        float(beta1), float(beta2), float(eps), float(weight_decay),
        _ptr(grad_scale), _ptr(found_inf), _ptr(skipped), _stream()),
        "ucsa_adam_step_scaled")


def adam_count_skipped(found_inf: torch.Tensor, skipped: torch.Tensor):
    check(lib().ucsa_adam_count_skipped(_ptr(found_inf), _ptr(skipped),
                                        _stream()), "ucsa_adam_count_skipped")


# ===================== losses / post-processing / metric ====================
def nerf_loss(rgb, sem, depth, gt_rgb, labels, gt_depth, uom: float,
              w_sem: float = 0.04, w_depth: float = 0.1,
              grad_scale: float = 1.0, want_grad: bool = True):
    """-> stats[8] (device), (d_rgb, d_sem, d_depth) | None."""
    rgb = _f32(rgb, "rgb").view(-1, 3)
    N = rgb.shape[0]
    sem = _f32(sem, "sem").view(N, -1)
    Cn = sem.shape[1]
    depth = _f32(depth, "depth").view(N)
    gt_rgb = _f32(gt_rgb, "gt_rgb").view(N, 3)
    gt_depth = _f32(gt_depth, "gt_depth").view(N)
    labels = labels.reshape(N).to(torch.int64).contiguous()
    dev = rgb.device
    stats = torch.empty(8, device=dev)
    partial = torch.empty(int(lib().ucsa_loss_partial_floats(N)), device=dev)
    grads = None
    if want_grad:
        grads = (torch.empty_like(rgb), torch.empty_like(sem),
                 torch.empty_like(depth))
    check(lib().ucsa_nerf_loss(
        _ptr(rgb), _ptr(sem), _ptr(depth), _ptr(gt_rgb), _ptr(labels),
        _ptr(gt_depth), N, Cn, float(uom), float(w_sem), float(w_depth),
        float(grad_scale), _ptr(stats),
        _ptr(grads[0]) if grads else None, _ptr(grads[1]) if grads else None,
        _ptr(grads[2]) if grads else None, _ptr(partial), _stream()),
        "ucsa_nerf_loss")
    return stats, grads


def nerf_loss_apply(grads, g_total, g_color, g_sem, g_depth, w_sem: float,
                    w_depth: float):
    """``grads`` = (d_rgb [N,3], d_sem [N,C], d_depth [N]) of ``nerf_loss``
    times the loss node's cotangents (0-d / 1-element device tensors or None),
    one launch (ucsa_nerf_loss_apply) into FRESH tensors: ``grads`` are the
    tensors the autograd node saved and a second backward through a retained
    graph must find them unscaled (the in-place form scaled them
    twice, unseen by autograd's version counter)."""
    d_rgb, d_sem, d_depth = grads
    N, Cn = d_sem.shape
    out = torch.empty(N * (4 + Cn), dtype=torch.float32, device=d_rgb.device)
    o_rgb = out[:N * 3].view(N, 3)
    o_sem = out[N * 3:N * (3 + Cn)].view(N, Cn)
    o_depth = out[N * (3 + Cn):]

    def sc(t):
        if t is None:
            return None
        t = t.detach().reshape(-1)
        if not t.is_cuda or t.dtype != torch.float32 or t.numel() != 1:
            t = t.to(d_rgb.device, torch.float32).reshape(1)
        return t.contiguous()

    gt, gc, gs, gd = sc(g_total), sc(g_color), sc(g_sem), sc(g_depth)
    check(lib().ucsa_nerf_loss_apply(_ptr(d_rgb), _ptr(d_sem), _ptr(d_depth),
                                     _ptr(o_rgb), _ptr(o_sem), _ptr(o_depth), N, Cn,
                                     _ptr(gt), _ptr(gc), _ptr(gs), _ptr(gd),
                                     float(w_sem), float(w_depth), _stream()),
          "ucsa_nerf_loss_apply")
    return o_rgb, o_sem, o_depth


def semantic_postproc(sem, want_normalised: bool = True):
    """[..., C] -> (normalised [..., C] | None, argmax [...] int64)."""
    shape = sem.shape
    s = _f32(sem, "sem").view(-1, shape[-1])
    N, Cn = s.shape
    norm = torch.empty_like(s) if want_normalised else None
    arg = torch.empty(N, dtype=torch.int64, device=s.device)
    check(lib().ucsa_semantic_postproc(_ptr(s), N, Cn, _ptr(norm), _ptr(arg),
                                       _stream()), "ucsa_semantic_postproc")
    return (None if norm is None else norm.view(shape)), arg.view(shape[:-1])


def seg_tail(logits, labels=None, grad_scale: float = 1.0,
             want_prob: bool = True, want_grad: bool = False):
    """logits [B,C,H,W] -> dict(prob, argmax, loss, d_logits)."""
    x = _f32(logits, "logits")
    B, Cn, H, W = x.shape
    P = H * W
    dev = x.device
    prob = torch.empty_like(x) if want_prob else None
    arg = torch.empty(B, H, W, dtype=torch.int64, device=dev)
    loss = d_logits = partial = None
    if labels is not None:
        labels = labels.reshape(B, P).to(torch.int64).contiguous()
        loss = torch.empty(1, device=dev)
        partial = torch.empty(int(lib().ucsa_loss_partial_floats(B * P)),
                              device=dev)
        if want_grad:
            d_logits = torch.empty_like(x)
    check(lib().ucsa_seg_tail(_ptr(x), _ptr(labels), B, Cn, P,
                              float(grad_scale), _ptr(prob), _ptr(arg),
                              _ptr(loss), _ptr(d_logits), _ptr(partial),
                              _stream()), "ucsa_seg_tail")
    return dict(prob=prob, argmax=arg, loss=loss, d_logits=d_logits)


# ---------------------------------------------------------------------------
# fused BatchNorm2d (+ residual) (+ ReLU) on channels-last activations
# ---------------------------------------------------------------------------
def _nhwc_rows(t: torch.Tensor, name: str):
    """[N,C,H,W] channels_last tensor (fp32 / bf16) -> (M, C, dtype code)."""
    if not t.is_cuda:
        raise _lib.UcsaError(f"{name} must live on the GPU: no CPU fallback")
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise _lib.UcsaError(f"{name} must be a channels_last [N,C,H,W] tensor")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.UcsaError(f"{name}: fp32 or bf16, got {t.dtype}")
    N, Cc, H, W = t.shape
    return N * H * W, Cc, 0 if t.dtype == torch.float32 else 1


BN_MAX_CHANNELS = 4096   # csrc/batchnorm.hip: per-workgroup column tables


def _bn_vec(t, name: str, Cc: int, device):
    """A per-channel operand of the BatchNorm kernels, which read ``float*`` of
    length C: fp32, contiguous, on x's device -- or None.  (A ``model.half()``
    / ``.bfloat16()`` module has 2-byte parameters: handing those over would be
    an out-of-bounds read and silently wrong statistics.)"""
    if t is None:
        return None
    if (t.dtype != torch.float32 or t.device != device or not t.is_contiguous()
            or t.numel() != Cc):
        raise _lib.UcsaError(
            f"bn_act: {name} must be a contiguous fp32 [{Cc}] tensor on {device}, got "
            f"{t.dtype} {tuple(t.shape)} on {t.device}")
    return t


def bn_vec_ok(t, Cc: int, device) -> bool:
    return t is None or (t.dtype == torch.float32 and t.device == device
                         and t.is_contiguous() and t.numel() == Cc)


def bn_act_fwd(x, residual, weight, bias, running_mean, running_var,
               momentum: float, eps: float, relu: bool, training: bool):
    """-> (y, save_mean, save_invstd); save_* are None in eval mode."""
    M, Cc, dt = _nhwc_rows(x, "x")
    if Cc > BN_MAX_CHANNELS:
        raise _lib.UcsaError(f"bn_act: C = {Cc} > {BN_MAX_CHANNELS}")
    for name, v in (("weight", weight), ("bias", bias), ("running_mean", running_mean),
                    ("running_var", running_var)):
        _bn_vec(v, name, Cc, x.device)
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.UcsaError("residual must match x (shape, dtype, channels_last)")
    y = torch.empty_like(x)   # preserves channels_last
    save_mean = save_invstd = None
    if training:
        save_mean = torch.empty(Cc, device=x.device)
        save_invstd = torch.empty(Cc, device=x.device)
    ws = _scratch(int(lib().ucsa_bn_workspace_bytes(M, Cc)), x.device)
    check(lib().ucsa_bn_act_fwd(
        _ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(running_mean),
        _ptr(running_var), float(momentum), float(eps), M, Cc, int(relu),
        int(training), dt, _ptr(y), _ptr(save_mean), _ptr(save_invstd), _ptr(ws),
        _stream()), "ucsa_bn_act_fwd")
    return y, save_mean, save_invstd


def bn_act_bwd(dy, x, y, weight, save_mean, save_invstd, relu: bool,
               want_dres: bool, want_dwb: bool):
    """-> (dx, dresidual | None, dweight | None, dbias | None)."""
    M, Cc, dt = _nhwc_rows(x, "x")
    if dy.dtype != x.dtype or dy.shape != x.shape:
        raise _lib.UcsaError("dy must match x")
    for name, v in (("weight", weight), ("save_mean", save_mean), ("save_invstd", save_invstd)):
        _bn_vec(v, name, Cc, x.device)
    if relu and (y is None or y.dtype != x.dtype or y.shape != x.shape or
                 not y.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.UcsaError("bn_act_bwd: y must match x (relu mask)")
    dy = dy.contiguous(memory_format=torch.channels_last)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dw = torch.empty(Cc, device=x.device) if want_dwb else None
    db = torch.empty(Cc, device=x.device) if want_dwb else None
    ws = _scratch(int(lib().ucsa_bn_workspace_bytes(M, Cc)), x.device)
    check(lib().ucsa_bn_act_bwd(
        _ptr(dy), _ptr(x), _ptr(y) if relu else None, _ptr(weight), _ptr(save_mean),
        _ptr(save_invstd), M, Cc, int(relu), dt, _ptr(dx), _ptr(dres), _ptr(dw),
        _ptr(db), _ptr(ws), _stream()), "ucsa_bn_act_bwd")
    return dx, dres, dw, db


def confusion_matrix(preds, truths, n_classes: int, cm=None):
    p = preds.reshape(-1).to(torch.int64).contiguous()
    t = truths.reshape(-1).to(torch.int64).contiguous()
    if not p.is_cuda:
        raise _lib.UcsaError("confusion_matrix needs GPU tensors")
    if cm is None:
        cm = torch.zeros(n_classes, n_classes, dtype=torch.int64,
                         device=p.device)
    check(lib().ucsa_confusion_matrix(_ptr(p), _ptr(t), p.numel(), n_classes,
                                      _ptr(cm), _stream()),
          "ucsa_confusion_matrix")
    return cm


# ---------------------------------------------------------------------------
# occupancy-grid ray marching
# ---------------------------------------------------------------------------
def _i32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.UcsaError(
            f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise _lib.UcsaError(f"{name} must be a contiguous int32 tensor")
    return t


def _inplace_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.UcsaError(
            f"{name} is updated in place: contiguous fp32 on the GPU required")
    return t


_march_ws = {}
_named_ws = {}


def _scratch_named(name: str, nbytes: int, device) -> torch.Tensor:
    """A grow-only scratch buffer of its own (not shared with _scratch's).
    Keyed by the calling THREAD as well as device and stream: the
    multi-launch ops (BatchNorm: stats -> finalize -> apply; augmentation: grey
    sums -> apply) pass intermediates through it, ctypes releases the GIL, and the
    trainer's prefetch thread issues on the same stream as the training step -- two
    threads sharing one buffer would overwrite each other's partial sums."""
    key = (name, device, torch.cuda.current_stream().cuda_stream, threading.get_ident())
    buf = _named_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _named_ws[key] = buf
    return buf


def _scratch(nbytes: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream, threading.get_ident())
    buf = _march_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8,
                          device=device)
        _march_ws[key] = buf
    return buf


def march_rays_train(rays_o, rays_d, density_grid, mean_density: float,
                     bound: float, dt_gamma: float, M: int, nears, fars, xyzs,
                     dirs, deltas, rays, counter, perturb: int):
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    grid = _f32(density_grid, "density_grid")
    N, Cc, H = rays_o.shape[0], grid.shape[0], grid.shape[1]
    ws = _scratch(int(lib().ucsa_march_workspace_bytes(N)), rays_o.device)
    check(lib().ucsa_march_rays_train(
        _ptr(rays_o), _ptr(rays_d), _ptr(grid), mean_density, bound, dt_gamma,
        N, Cc, H, M, _ptr(_f32(nears, "nears")), _ptr(_f32(fars, "fars")),
        _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(_i32(rays, "rays")),
        _ptr(_i32(counter, "step_counter")), int(perturb), _ptr(ws),
        _stream()), "ucsa_march_rays_train")


def composite_rays_train_fwd(sigmas, rgbs, local_sem, deltas, rays):
    sigmas = _f32(sigmas, "sigmas").view(-1)
    M, N = sigmas.shape[0], rays.shape[0]
    rgbs = _f32(rgbs, "rgbs").view(M, 3)
    deltas = _f32(deltas, "deltas").view(M, 2)
    dev = sigmas.device
    n_sem = 0
    if local_sem is not None:
        local_sem = _f32(local_sem, "local_semantics").view(M, -1)
        n_sem = local_sem.shape[1]
    ws = torch.empty(N, device=dev)
    depth = torch.empty(N, device=dev)
    image = torch.empty(N, 3, device=dev)
    sem = torch.empty(N, n_sem, device=dev) if n_sem else None
    check(lib().ucsa_composite_rays_train_fwd(
        _ptr(sigmas), _ptr(rgbs), _ptr(local_sem), _ptr(deltas),
        _ptr(_i32(rays, "rays")), M, N, n_sem, _ptr(ws), _ptr(depth),
        _ptr(image), _ptr(sem), _stream()), "ucsa_composite_rays_train_fwd")
    return ws, depth, image, sem


def composite_rays_train_bwd(grad_ws, grad_image, grad_sem, sigmas, rgbs,
                             deltas, rays, weights_sum, image):
    M, N = sigmas.shape[0], rays.shape[0]
    dev = sigmas.device
    n_sem = 0 if grad_sem is None else grad_sem.shape[1]
    g_sig = torch.zeros(M, device=dev)
    g_rgb = torch.zeros(M, 3, device=dev)
    g_ls = torch.zeros(M, n_sem, device=dev) if n_sem else None
    check(lib().ucsa_composite_rays_train_bwd(
        _ptr(_f32(grad_ws, "grad_weights_sum")),
        _ptr(_f32(grad_image, "grad_image")),
        _ptr(None if grad_sem is None else _f32(grad_sem, "grad_semantics")),
        _ptr(sigmas), _ptr(rgbs), _ptr(deltas), _ptr(rays), _ptr(weights_sum),
        _ptr(image), M, N, n_sem, _ptr(g_sig), _ptr(g_rgb), _ptr(g_ls),
        _stream()), "ucsa_composite_rays_train_bwd")
    return g_sig, g_rgb, g_ls


def march_rays(n_alive: int, n_step: int, rays_alive, rays_t, rays_o, rays_d,
               bound: float, dt_gamma: float, density_grid, mean_density: float,
               nears, fars, xyzs, dirs, deltas, perturb: int):
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    grid = _f32(density_grid, "density_grid")
    check(lib().ucsa_march_rays(
        n_alive, n_step, _ptr(_i32(rays_alive, "rays_alive")),
        _ptr(_f32(rays_t, "rays_t")), _ptr(rays_o), _ptr(rays_d), bound,
        dt_gamma, grid.shape[0], grid.shape[1], _ptr(grid), mean_density,
        _ptr(_f32(nears, "nears")), _ptr(_f32(fars, "fars")), _ptr(xyzs),
        _ptr(dirs), _ptr(deltas), int(perturb), _stream()), "ucsa_march_rays")


def composite_rays(n_alive: int, n_step: int, rays_alive, rays_t, sigmas, rgbs,
                   local_sem, deltas, weights_sum, depth, image, semantics):
    n_sem = 0
    if local_sem is not None:
        local_sem = _f32(local_sem, "local_semantics")
        n_sem = local_sem.shape[-1]
        _inplace_f32(semantics, "semantics")
    check(lib().ucsa_composite_rays(
        n_alive, n_step, _ptr(_i32(rays_alive, "rays_alive")),
        _ptr(_inplace_f32(rays_t, "rays_t")), _ptr(_f32(sigmas, "sigmas")),
        _ptr(_f32(rgbs, "rgbs")), _ptr(local_sem), _ptr(_f32(deltas, "deltas")),
        n_sem, _ptr(_inplace_f32(weights_sum, "weights_sum")),
        _ptr(_inplace_f32(depth, "depth")), _ptr(_inplace_f32(image, "image")),
        _ptr(semantics), _stream()), "ucsa_composite_rays")


def compact_rays(n_alive: int, rays_alive, rays_alive_old, rays_t, rays_t_old,
                 alive_counter):
    ws = _scratch(int(lib().ucsa_compact_workspace_bytes(n_alive)),
                  rays_t.device)
    check(lib().ucsa_compact_rays(
        n_alive, _ptr(_i32(rays_alive, "rays_alive")),
        _ptr(_i32(rays_alive_old, "rays_alive_old")),
        _ptr(_inplace_f32(rays_t, "rays_t")),
        _ptr(_f32(rays_t_old, "rays_t_old")),
        _ptr(_i32(alive_counter, "alive_counter")), _ptr(ws), _stream()),
        "ucsa_compact_rays")


def density_grid_points(cascade: int, H: int, bound: float, seed: int, device):
    xyz = torch.empty(H * H * H, 3, device=device)
    check(lib().ucsa_density_grid_points(cascade, H, bound, seed, _ptr(xyz),
                                         _stream()), "ucsa_density_grid_points")
    return xyz


def density_grid_update(density_grid, fresh, decay: float, fresh_scale: float):
    """In place on density_grid; returns the new mean density (device [1])."""
    _inplace_f32(density_grid, "density_grid")
    fresh = _f32(fresh, "fresh")
    n = density_grid.numel()
    mean = torch.empty(1, device=density_grid.device)
    ws = _scratch(int(lib().ucsa_density_grid_workspace_bytes()),
                  density_grid.device)
    check(lib().ucsa_density_grid_update(_ptr(density_grid), _ptr(fresh), n,
                                         decay, fresh_scale, _ptr(mean),
                                         _ptr(ws), _stream()),
          "ucsa_density_grid_update")
    return mean


class MarchSegments:
    """Buffers + calls of the segmented marcher (``ucsa_march_segment_*``) for
    one batch of N rays; see include/ucsa_hip.h."""

    def __init__(self, rays_o, rays_d, nears, fars, density_grid,
                 mean_density: float, bound: float, dt_gamma: float):
        self.o = _f32(rays_o, "rays_o").view(-1, 3)
        self.d = _f32(rays_d, "rays_d").view(-1, 3)
        self.fars = _f32(fars, "fars")
        self.grid = _f32(density_grid, "density_grid")
        self.args = (float(bound), float(dt_gamma), self.grid.shape[0],
                     self.grid.shape[1])
        self.mean_density = float(mean_density)
        N = self.o.shape[0]
        dev = self.o.device
        self.N = N
        self.alive = torch.empty(2, N, dtype=torch.int32, device=dev)
        self.alive[0] = torch.arange(N, dtype=torch.int32, device=dev)
        self.t = torch.empty(2, N, device=dev)
        self.t[0] = _f32(nears, "nears")
        self.span = torch.empty(N, 2, dtype=torch.int32, device=dev)
        self.n_alive = torch.empty(2, dtype=torch.int32, device=dev)
        self.ws = torch.empty(
            int(lib().ucsa_march_segment_workspace_bytes(N)) // 4,
            dtype=torch.int32, device=dev)
        self.cur = 0        # which half of alive / t / n_alive is current
        self.first = True   # round 0: every ray alive, count not on device yet
        self.stage = None   # staging rows of the current round (or None)
        self.stage_cap = 0
        # rounds whose staging rows fit this many bytes march once (the
        # samples are staged while counting, then copied into place)
        self.stage_limit = 1 << 30

    def _n_dev(self):
        return None if self.first else _ptr(self.n_alive[self.cur:])

    def count(self, n_cap: int, cap: int, perturb: int):
        b, g, C_, H = self.args
        need = int(lib().ucsa_march_segment_stage_bytes(n_cap, cap))
        self.stage, self.stage_cap = None, cap
        if 0 < need <= self.stage_limit:
            self.stage = _scratch_named("march_stage", need, self.o.device)
        check(lib().ucsa_march_segment_count(
            n_cap, self._n_dev(), cap, _ptr(self.alive[self.cur]),
            _ptr(self.t[self.cur]), _ptr(self.o), _ptr(self.d), b, g, C_, H,
            _ptr(self.grid), self.mean_density, _ptr(self.fars), int(perturb),
            _ptr(self.span), _ptr(self.ws), _ptr(self.stage), _stream()),
            "ucsa_march_segment_count")
        total, n_alive = self.ws[:2].tolist()   # the round's one host sync
        return total, n_alive

    def write(self, n_cap: int, M: int, perturb: int):
        b, g, C_, H = self.args
        dev = self.o.device
        xyzs = torch.empty(M, 3, device=dev)
        dirs = torch.empty(M, 3, device=dev)
        deltas = torch.empty(M, 2, device=dev)
        check(lib().ucsa_march_segment_write(
            n_cap, self._n_dev(), _ptr(self.alive[self.cur]),
            _ptr(self.t[self.cur]), _ptr(self.o), _ptr(self.d), b, g, C_, H,
            _ptr(self.grid), self.mean_density, _ptr(self.fars), int(perturb),
            _ptr(self.span), _ptr(xyzs), _ptr(dirs), _ptr(deltas),
            _ptr(self.stage), self.stage_cap, _stream()),
            "ucsa_march_segment_write")
        return xyzs, dirs, deltas

    def composite(self, n_cap: int, cap: int, sigmas, sigma_scale: float, rgbs,
                  local_sem, deltas, weights_sum, depth, image, semantics):
        n_sem = 0 if local_sem is None else local_sem.shape[-1]
        check(lib().ucsa_march_segment_composite(
            n_cap, self._n_dev(), cap, _ptr(self.alive[self.cur]),
            _ptr(self.t[self.cur]), _ptr(self.span), _ptr(sigmas),
            float(sigma_scale), _ptr(rgbs), _ptr(local_sem), _ptr(deltas),
            n_sem, _ptr(weights_sum), _ptr(depth), _ptr(image),
            _ptr(semantics), _stream()), "ucsa_march_segment_composite")

    def shade(self, n_cap: int, cap: int, sigmas, sigma_scale: float, h,
              deltas, packed_color, packed_sem, n_classes: int, w_min: float,
              weights_sum, depth, image, semantics, half=False):
        """``half``: False -- f32-input MFMA nets (packed by mlp_pack); True / "fp16" --
        plain f16 nets (mlp_pack_f16); "f16x2" -- two-term f16 nets (mlp_pack_h2)."""
        fn = (lib().ucsa_march_segment_shade_h2 if half == "f16x2"
              else lib().ucsa_march_segment_shade_f16 if half
              else lib().ucsa_march_segment_shade)
        check(fn(n_cap, self._n_dev(), cap, _ptr(self.alive[self.cur]),
                 _ptr(self.t[self.cur]), _ptr(self.span), _ptr(self.d),
                 _ptr(sigmas), float(sigma_scale), _ptr(h), _ptr(deltas),
                 _ptr(packed_color), _ptr(packed_sem), n_classes, float(w_min),
                 _ptr(weights_sum), _ptr(depth), _ptr(image), _ptr(semantics),
                 _stream()), "ucsa_march_segment_shade")

    def compact(self, n_cap: int):
        nxt = 1 - self.cur
        check(lib().ucsa_march_segment_compact(
            n_cap, self._n_dev(), _ptr(self.alive[nxt]),
            _ptr(self.alive[self.cur]), _ptr(self.t[nxt]),
            _ptr(self.t[self.cur]), _ptr(self.n_alive[nxt:]), _ptr(self.ws),
            _stream()), "ucsa_march_segment_compact")
        self.cur = nxt
        self.first = False


def march_train_fwd(rays, M: int, nears, rays_d, sigmas, sigma_scale: float, h,
                    deltas, packed_color, packed_sem, n_classes: int,
                    w_min: float):
    """-> weights_sum [N], depth_raw [N] (sum w*t), image [N,3], sem [N,C],
    w [M], t [M]."""
    N = rays.shape[0]
    dev = rays.device
    ws = torch.zeros(N, device=dev)
    depth = torch.zeros(N, device=dev)
    image = torch.zeros(N, 3, device=dev)
    sem = torch.zeros(N, n_classes, device=dev)
    w = torch.zeros(M, device=dev)
    t = torch.zeros(M, device=dev)
    check(lib().ucsa_march_train_fwd(
        _ptr(_i32(rays, "rays")), N, M, _ptr(_f32(nears, "nears")),
        _ptr(_f32(rays_d, "rays_d")), _ptr(sigmas), float(sigma_scale), _ptr(h),
        _ptr(deltas), _ptr(packed_color), _ptr(packed_sem), n_classes,
        float(w_min), _ptr(ws), _ptr(depth), _ptr(image), _ptr(sem), _ptr(w),
        _ptr(t), _stream()), "ucsa_march_train_fwd")
    return ws, depth, image, sem, w, t


def march_train_bwd(rays, M: int, rays_d, norms, sigmas, sigma_scale: float, h,
                    deltas, w, t, packed_color, packed_sem, packed_color_t,
                    packed_sem_t, n_classes: int, w_min: float, d_image,
                    d_depth, d_sem):
    """-> d_h [M,16], partial_color, partial_sem."""
    N = rays.shape[0]
    dev = rays.device
    G = torch.empty(max(M, 1), device=dev)
    d_h = torch.empty(max(M, 1), 16, device=dev)
    parts = int(lib().ucsa_composite_bwd_parts(N))
    nrb = (n_classes + 15) // 16
    pc = torch.empty(parts, 7168, device=dev)
    ps = torch.empty(parts, 1024 + 1024 * nrb, device=dev)
    check(lib().ucsa_march_train_bwd(
        _ptr(rays), N, M, _ptr(rays_d), _ptr(norms), _ptr(sigmas),
        float(sigma_scale), _ptr(h), _ptr(deltas), _ptr(w), _ptr(t),
        _ptr(packed_color), _ptr(packed_sem), _ptr(packed_color_t),
        _ptr(packed_sem_t), n_classes, float(w_min),
        _ptr(_f32(d_image, "d_image")), _ptr(_f32(d_depth, "d_depth")),
        _ptr(_f32(d_sem, "d_sem")), _ptr(G), _ptr(d_h), _ptr(pc), _ptr(ps),
        _stream()), "ucsa_march_train_bwd")
    return d_h[:M], pc, ps


def augment(img, label, params, out_size=None):
    """Rendered-image augmentation (ucsa_augment).  img [B,3,H,W] fp32 in
    [0,1], label [B,H,W] int64 or None, params: list of B dicts with keys
    order (4 ints), brightness, contrast, saturation, hue, angle_deg, flip,
    crop_i, crop_j.  -> out_img [B,3,oh,ow], out_label [B,oh,ow] | None."""
    img = _f32(img, "img")
    B, _, H, W = img.shape
    oh, ow = (H, W) if out_size is None else out_size
    if label is not None:
        if not (label.is_cuda and label.dtype == torch.int64):
            raise _lib.UcsaError("label must be an int64 tensor on the GPU")
        label = label.contiguous()
    arr = (_lib.AugParams * B)()
    for k, q in enumerate(params):
        arr[k].order[:] = [int(v) for v in q["order"]]
        arr[k].brightness = float(q["brightness"])
        arr[k].contrast = float(q["contrast"])
        arr[k].saturation = float(q["saturation"])
        arr[k].hue = float(q["hue"])
        arr[k].angle_deg = float(q["angle_deg"])
        arr[k].flip = int(bool(q["flip"]))
        arr[k].crop_i = int(q.get("crop_i", 0))
        arr[k].crop_j = int(q.get("crop_j", 0))
    out = torch.empty(B, 3, oh, ow, device=img.device)
    out_l = (torch.empty(B, oh, ow, dtype=torch.int64, device=img.device)
             if label is not None else None)
    ws = _scratch(int(lib().ucsa_augment_workspace_bytes(B, H, W)), img.device)
    check(lib().ucsa_augment(_ptr(img), _ptr(label), B, H, W, arr, oh, ow,
                             _ptr(out), _ptr(out_l), _ptr(ws), _stream()),
          "ucsa_augment")
    return out, out_l

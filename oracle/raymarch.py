"""numpy front end of ``oracle/raymarch.c`` -- TEST INFRASTRUCTURE, NOT PRODUCT.

The C file restates the reference's occupancy-grid marching kernels
(``nr4seg/nerf/raymarching/src/raymarching.cu:138-855``); this module compiles
it with gcc on first use (``make -C oracle``) and mirrors the argument
conventions of the reference wrappers (``raymarching.py:54-595``) on numpy
arrays.  PARITY UNPINNED (CUDA-only reference, no vectors): see the header of
``raymarch.c`` for what pins it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraymarch_oracle.so")
_SRC = os.path.join(_HERE, "raymarch.c")
_lib = None

MAX_STEPS = 1024
MIN_STEPSIZE = np.float32(2) * np.float32(1.73205080757) / np.float32(1024)


def build() -> str:
    if (not os.path.exists(_SO)
            or os.path.getmtime(_SO) < os.path.getmtime(_SRC)):
        res = subprocess.run(["make", "-C", _HERE], capture_output=True,
                             text=True)
        if res.returncode != 0:
            raise RuntimeError("building the C oracle failed:\n" + res.stdout +
                               res.stderr)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_pcg32_first_float.restype = C.c_float
        _lib.orc_pcg32_first_float.argtypes = [C.c_uint64, C.c_uint64]
        _lib.orc_pcg32_sequence.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32,
                                            C.c_void_p]
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def pcg32_sequence(initstate: int, initseq: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint32)
    lib().orc_pcg32_sequence(initstate, initseq, n, _p(out))
    return out


def pcg32_first_float(initstate: int, initseq: int = 1) -> float:
    return float(lib().orc_pcg32_first_float(initstate, initseq))


def march_rays_train(rays_o, rays_d, bound, density_grid, mean_density, nears,
                     fars, step_counter=None, mean_count=-1, perturb=False,
                     align=-1, force_all_rays=False, dt_gamma=0.0):
    """reference raymarching.py:54-163 -> xyzs, dirs, deltas, rays (+ the
    counter, returned last)."""
    rays_o, rays_d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    grid = _f(density_grid)
    N, Cc, H = rays_o.shape[0], grid.shape[0], grid.shape[1]
    M = N * 1024
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    if step_counter is None:
        step_counter = np.zeros(2, np.int32)
    lib().orc_march_rays_train(
        _p(rays_o), _p(rays_d), _p(grid), C.c_float(mean_density),
        C.c_float(bound), C.c_float(dt_gamma), C.c_uint32(N), C.c_uint32(Cc),
        C.c_uint32(H), C.c_uint32(M), _p(_f(nears)), _p(_f(fars)), _p(xyzs),
        _p(dirs), _p(deltas), _p(rays), _p(step_counter),
        C.c_uint32(int(perturb)))
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays, step_counter


def composite_rays_train(sigmas, rgbs, deltas, rays, local_semantics=None):
    """reference raymarching.py:169-203 (and :249-309 with semantics) ->
    weights_sum [N], depth [N], image [N,3] (, semantics [N,Cs])."""
    sigmas, rgbs, deltas = _f(sigmas), _f(rgbs), _f(deltas)
    rays = _i(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ls = None if local_semantics is None else _f(local_semantics)
    n_sem = 0 if ls is None else ls.shape[1]
    ws = np.zeros(N, np.float32)
    depth = np.zeros(N, np.float32)
    image = np.zeros((N, 3), np.float32)
    sem = np.zeros((N, n_sem), np.float32) if n_sem else None
    lib().orc_composite_rays_train_fwd(
        _p(sigmas), _p(rgbs), _p(ls), _p(deltas), _p(rays), C.c_uint32(M),
        C.c_uint32(N), C.c_uint32(n_sem), _p(ws), _p(depth), _p(image),
        _p(sem))
    return (ws, depth, image) if sem is None else (ws, depth, image, sem)


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas,
                                  rays, weights_sum, image, grad_sem=None):
    """reference raymarching.py:205-243 -> grad_sigmas [M], grad_rgbs [M,3]
    (, grad_local_semantics [M,Cs])."""
    sigmas, rgbs, deltas = _f(sigmas), _f(rgbs), _f(deltas)
    rays = _i(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    gs = None if grad_sem is None else _f(grad_sem)
    n_sem = 0 if gs is None else gs.shape[1]
    g_sig = np.zeros(M, np.float32)
    g_rgb = np.zeros((M, 3), np.float32)
    g_ls = np.zeros((M, n_sem), np.float32) if n_sem else None
    lib().orc_composite_rays_train_bwd(
        _p(_f(grad_ws)), _p(_f(grad_image)), _p(gs), _p(sigmas), _p(rgbs),
        _p(deltas), _p(rays), _p(_f(weights_sum)), _p(_f(image)),
        C.c_uint32(M), C.c_uint32(N), C.c_uint32(n_sem), _p(g_sig), _p(g_rgb),
        _p(g_ls))
    return (g_sig, g_rgb) if g_ls is None else (g_sig, g_rgb, g_ls)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
               density_grid, mean_density, near, far, align=-1, perturb=False,
               dt_gamma=0.0):
    """reference raymarching.py:367-449 -> xyzs, dirs, deltas."""
    rays_o, rays_d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    grid = _f(density_grid)
    Cc, H = grid.shape[0], grid.shape[1]
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(
        C.c_uint32(n_alive), C.c_uint32(n_step), _p(_i(rays_alive)),
        _p(_f(rays_t)), _p(rays_o), _p(rays_d), C.c_float(bound),
        C.c_float(dt_gamma), C.c_uint32(Cc), C.c_uint32(H), _p(grid),
        C.c_float(mean_density), _p(_f(near)), _p(_f(far)), _p(xyzs),
        _p(dirs), _p(deltas), C.c_uint32(int(perturb)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas,
                   weights_sum, depth, image, local_semantics=None,
                   semantics=None):
    """reference raymarching.py:455-501 / :507-555; in place on rays_t,
    weights_sum, depth, image (, semantics) which must be contiguous fp32."""
    for a in (rays_t, weights_sum, depth, image):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    ls = None if local_semantics is None else _f(local_semantics)
    n_sem = 0 if ls is None else ls.shape[1]
    if n_sem:
        assert semantics.dtype == np.float32 and semantics.flags.c_contiguous
    lib().orc_composite_rays(
        C.c_uint32(n_alive), C.c_uint32(n_step), _p(_i(rays_alive)),
        _p(rays_t), _p(_f(sigmas)), _p(_f(rgbs)), _p(ls), _p(_f(deltas)),
        C.c_uint32(n_sem), _p(weights_sum), _p(depth), _p(image),
        _p(semantics))


def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old,
                 alive_counter):
    """reference raymarching.py:561-592; in place on rays_alive, rays_t,
    alive_counter (int32 / fp32 contiguous)."""
    assert rays_alive.dtype == np.int32 and rays_t.dtype == np.float32
    assert alive_counter.dtype == np.int32
    lib().orc_compact_rays(C.c_uint32(n_alive), _p(rays_alive),
                           _p(_i(rays_alive_old)), _p(rays_t),
                           _p(_f(rays_t_old)), _p(alive_counter))

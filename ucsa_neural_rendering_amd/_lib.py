"""ctypes binding of ``libucsa_hip.so`` (the C ABI declared in
``include/ucsa_hip.h``).

The library is the product: there is no CPU or PyTorch fallback.  If the
shared object is missing or a symbol is absent, importing/using the ops fails
loudly with instructions to build it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

# PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and
# must be loaded BEFORE libucsa_hip.so so that both resolve to one HIP runtime
# instance; the other order maps a second runtime that sees no device.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libucsa_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MAX_LEVELS = 16
MLP_SIGMA, MLP_COLOR, MLP_SEM = 0, 1, 2


class GridLevel(C.Structure):
    _fields_ = [("scale", C.c_float), ("res", C.c_uint32),
                ("entries", C.c_uint32), ("offset", C.c_uint32),
                ("hashed", C.c_uint32)]


class Grid(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32),
                ("total_entries", C.c_uint32), ("bound", C.c_float),
                ("level", GridLevel * MAX_LEVELS)]


class TrainBuffers(C.Structure):
    """ucsa_train_buffers: device pointers of the training forward's saved state."""
    _fields_ = [(k, C.c_void_p) for k in ("z_c", "feat_c", "h_c", "sigma_c", "z_f",
                                          "feat_f", "h_f", "sigma_f", "src", "weights")]


class TrainPacks(C.Structure):
    """ucsa_train_packs: bf16x3 weight fragments (forward) and their transposes."""
    _fields_ = [(k, C.c_void_p) for k in ("sigma_x3", "color_x3", "sem_x3",
                                          "sigma_t_x3", "color_t_x3", "sem_t_x3",
                                          "sigma_h2", "color_h2", "sem_h2")]


class AugParams(C.Structure):
    _fields_ = [("order", C.c_int32 * 4), ("brightness", C.c_float),
                ("contrast", C.c_float), ("saturation", C.c_float),
                ("hue", C.c_float), ("angle_deg", C.c_float),
                ("flip", C.c_int32), ("crop_i", C.c_int32),
                ("crop_j", C.c_int32)]


_p = C.c_void_p
_u32 = C.c_uint32
_f = C.c_float

# name -> (restype, argtypes); must list every symbol of include/ucsa_hip.h
SIGNATURES = {
    "ucsa_version": (C.c_int32, []),
    "ucsa_error_string": (C.c_char_p, [C.c_int32]),
    "ucsa_grid_init": (C.c_int32, [C.POINTER(Grid), _f, _u32, _u32, _u32,
                                   C.c_double]),
    "ucsa_get_rays": (C.c_int32, [_p, _u32, _f, _f, _f, _f, _u32, _u32, _p,
                                  _u32, _p, _p, _p, _p]),
    "ucsa_tile_order": (C.c_int32, [_p, _u32, _u32, _u32, _u32, _p, _p]),
    "ucsa_near_far_from_aabb": (C.c_int32, [_p, _p, C.POINTER(_f), _u32, _f,
                                            _p, _p, _p]),
    "ucsa_sample_coarse": (C.c_int32, [_p, _p, _p, _u32, _u32, _p, _p]),
    "ucsa_hashgrid_encode_rays": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p,
                                              C.POINTER(_f), _u32, _u32, _p,
                                              _p]),
    "ucsa_hashgrid_encode_rays_image": (C.c_int32, [C.POINTER(Grid), _p, _p, _p,
                                                    _p, C.POINTER(_f), _u32,
                                                    _u32, _u32, _p, _p]),
    "ucsa_hashgrid_encode_points": (C.c_int32, [C.POINTER(Grid), _p, _p, _u32,
                                                _p, _p]),
    "ucsa_env_reload": (None, []),
    "ucsa_mlp_pack": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_sigma_mlp_fwd": (C.c_int32, [_p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_resample": (C.c_int32, [_p, _p, _p, _u32, _u32, _u32, _f, _p, _p]),
    "ucsa_composite_fwd": (C.c_int32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                       _u32, _u32, _u32, _u32, _f, _p, _p, _p,
                                       _p, _p, _p]),
    "ucsa_composite_infer_workspace_bytes": (C.c_uint64, [_u32, _u32, _u32]),
    "ucsa_composite_infer": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32, _f,
                                                      _p, _p, _p, _p, _p]),
    "ucsa_composite_infer_f16": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32,
                                                          _f, _p, _p, _p, _p, _p]),
    "ucsa_render_workspace_bytes": (C.c_uint64, [_u32, _u32, _u32, _u32]),
    "ucsa_render_fwd": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p, _p,
                                    _p, C.POINTER(_f), _f, _p, _p, _u32, _u32,
                                    _u32, _u32, _f, _u32, _p, _p, _p, _p, _p]),
    "ucsa_point_shade": (C.c_int32, [_p, _p, _p, _p, _p, _u32, _u32, _p, _p,
                                     _p]),
    "ucsa_point_shade_h": (C.c_int32, [_p, _p, _p, _p, _p, _u32, _u32, _p, _p,
                                       _p]),
    # ---- fp16-MFMA inference option ----
    "ucsa_cast_f32_to_f16": (C.c_int32, [_p, _p, C.c_uint64, _p]),
    "ucsa_hashgrid_encode_rays_h16": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p,
                                                  C.POINTER(_f), _u32, _u32, _u32,
                                                  _p, _p]),
    "ucsa_hashgrid_encode_rays_hf": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p,
                                                 C.POINTER(_f), _u32, _u32, _u32,
                                                 _p, _p]),
    "ucsa_hashgrid_bwd_det_workspace_bytes": (C.c_uint64, [C.POINTER(Grid)]),
    "ucsa_hashgrid_bwd_rays_det": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, C.POINTER(_f),
                                               _u32, _u32, _p, _p, _p]),
    "ucsa_hashgrid_bwd_det_finish": (C.c_int32, [C.POINTER(Grid), _p, _p, _p]),
    "ucsa_tile_depth_order": (C.c_int32, [_p, _u32, _u32, _u32, _p, _p, _p, _p]),
    "ucsa_tile_index_order": (C.c_int32, [_p, _u32, _u32, _u32, _p, _p, _p, _p]),
    "ucsa_hashgrid_encode_sorted": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                                C.POINTER(_f), _u32, _u32, _u32,
                                                _p, _p]),
    "ucsa_hashgrid_encode_sorted_hf": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                                   C.POINTER(_f), _u32, _u32, _u32,
                                                   _p, _p]),
    "ucsa_density_sorted": (C.c_int32, [C.c_int32, C.POINTER(Grid), _p, _p, _p, _p, _p, _p,
                                        C.POINTER(_f), _u32, _u32, _u32, _p, _p, _p, _p,
                                        _p]),
    "ucsa_sigma_mlp_fwd_scatter": (C.c_int32, [C.c_int32, _p, _p, _u32, _u32, _p,
                                               _p, _p, _p]),
    "ucsa_sigma_mlp_fwd_f16_h": (C.c_int32, [_p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_render_fwd_f16_h16": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                            _p, _p, C.POINTER(_f), _f, _p, _p,
                                            _u32, _u32, _u32, _u32, _f, _u32, _p,
                                            _p, _p, _p, _p]),
    "ucsa_mlp_pack_h2_bytes": (C.c_uint32, [C.c_int32, _u32]),
    "ucsa_mlp_pack_h2": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_mlp_pack_h2_checked": (C.c_int32, [C.c_int32, _p, _p, _u32, _p, _p, _p]),
    "ucsa_sigma_mlp_fwd_h2": (C.c_int32, [_p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_render_fwd_h2": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                       _p, _p, C.POINTER(_f), _f, _p, _p,
                                       _u32, _u32, _u32, _u32, _f, _u32, _p,
                                       _p, _p, _p, _p]),
    "ucsa_composite_infer_h2": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32,
                                                         _f, _p, _p, _p, _p, _p]),
    "ucsa_composite_train_fwd_h2": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32,
                                                             _f, _p, _p, _p, _p, _p,
                                                             _p, _p]),
    "ucsa_mlp_pack_x3_bytes": (C.c_uint32, [C.c_int32, _u32]),
    "ucsa_mlp_pack_x3": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_sigma_mlp_fwd_x3": (C.c_int32, [_p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_render_view": (C.c_int32, [_u32, C.POINTER(Grid)] + [_p] * 7 + [C.POINTER(_f), _f, _p, _p,
                                     _u32, _u32, _u32, _u32, _f, _u32, _u32] + [_p] * 6),
    "ucsa_render_fwd_x3": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                       _p, _p, C.POINTER(_f), _f, _p, _p,
                                       _u32, _u32, _u32, _u32, _f, _u32, _p,
                                       _p, _p, _p, _p]),
    "ucsa_composite_infer_x3": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32,
                                                         _f, _p, _p, _p, _p, _p]),
    "ucsa_composite_train_fwd_x3": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32,
                                                             _f, _p, _p, _p, _p, _p,
                                                             _p, _p]),
    "ucsa_mlp_pack_f16_halves": (C.c_uint32, [C.c_int32, _u32]),
    "ucsa_mlp_pack_f16": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_sigma_mlp_fwd_f16": (C.c_int32, [_p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_composite_fwd_f16": (C.c_int32, [_p] * 10 + [_u32, _u32, _u32, _u32,
                                                        _f, _p, _p, _p, _p, _p,
                                                        _p]),
    "ucsa_render_fwd_f16": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                        _p, _p, C.POINTER(_f), _f, _p, _p,
                                        _u32, _u32, _u32, _u32, _f, _u32, _p,
                                        _p, _p, _p, _p]),
    # ---- training ----
    "ucsa_mlp_pack_t_size": (C.c_uint32, [C.c_int32, _u32]),
    "ucsa_mlp_pack_t": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_reduce_partials": (C.c_int32, [_p, _u32, _u32, C.c_int32, _p, _p]),
    "ucsa_reduce_partials_multi": (C.c_int32, [_u32, C.POINTER(_p),
                                               C.POINTER(_u32), C.POINTER(_u32),
                                               C.POINTER(_p), C.c_int32, _p]),
    "ucsa_sigma_mlp_bwd_parts": (C.c_uint32, [_u32]),
    "ucsa_sigma_mlp_bwd": (C.c_int32, [_p, _p, _p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_sigma_mlp_bwd_x2": (C.c_int32, [_p, _p, _p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_sigma_mlp_bwd_h16": (C.c_int32, [_p, _p, _p, _p, _u32, _u32, _p, _p, _p]),
    "ucsa_hashgrid_bwd_workspace_bytes": (C.c_uint64, [_u32, _u32, _u32]),
    "ucsa_hashgrid_bwd_rays": (C.c_int32, [C.POINTER(Grid), _p, _p, _p,
                                           C.POINTER(_f), _u32, _u32, _p, _p,
                                           _p, _p]),
    "ucsa_hashgrid_bwd_rays_h16": (C.c_int32, [C.POINTER(Grid), _p, _p, _p,
                                               C.POINTER(_f), _u32, _u32, _p, _p,
                                               _p, _f, _p]),
    "ucsa_hashgrid_bwd_rays_merged": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                                  C.POINTER(_f), _u32, _u32, _u32, _p, _p,
                                                  _p, _p, _p]),
    "ucsa_hashgrid_bwd_rays_p64": (C.c_int32, [C.POINTER(Grid), _p, _p, _p,
                                               C.POINTER(_f), _u32, _u32, _p, _p,
                                               _p, _p]),
    "ucsa_hashgrid_bwd_rays_merged_p64": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                                      C.POINTER(_f), _u32, _u32, _u32, _p, _p,
                                                      _p, _p, _p]),
    "ucsa_render_fused_fwd_workspace_bytes": (C.c_uint64, [_u32, _u32, _u32]),
    "ucsa_render_fused_fwd": (C.c_int32, [C.POINTER(Grid), _p, C.POINTER(TrainPacks), _p, _p, _p,
                                          C.POINTER(_f), _f, _p, _p, _u32, _u32, _u32, _u32, _f,
                                          C.POINTER(TrainBuffers), _p, _p, _p, _p, _p]),
    "ucsa_render_fused_bwd_workspace_bytes": (C.c_uint64, [_u32, _u32, _u32, _u32, _u32]),
    "ucsa_render_fused_bwd": (C.c_int32, [C.POINTER(Grid), C.POINTER(TrainPacks), _p, _p, _p,
                                          C.POINTER(_f), C.POINTER(TrainBuffers), _p, _p, _p,
                                          _u32, _u32, _u32, _u32, _f, _p, _p, _p, _p, _p, _p]),
    "ucsa_hashgrid_bwd_rays_merged_h16": (C.c_int32, [C.POINTER(Grid), _p, _p, _p, _p, _p,
                                                      C.POINTER(_f), _u32, _u32, _u32, _p, _p,
                                                      _p, _p, _f, _p]),
    "ucsa_hashgrid_bwd_points": (C.c_int32, [C.POINTER(Grid), _p, _u32, _p, _p,
                                             _p, _p]),
    "ucsa_composite_bwd_parts": (C.c_uint32, [_u32]),
    "ucsa_composite_bwd": (C.c_int32, [_p] * 17 + [_u32, _u32, _u32, _u32, _f] +
                           [_p] * 6),
    "ucsa_mlp_pack_t_f16_halves": (C.c_uint32, [C.c_int32, _u32]),
    "ucsa_mlp_pack_t_f16": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_composite_bwd_parts_f16": (C.c_uint32, [_u32]),
    "ucsa_mlp_pack_t_x3_bytes": (C.c_uint32, [C.c_int32, _u32]),
    "ucsa_mlp_pack_t_x3": (C.c_int32, [C.c_int32, _p, _p, _u32, _p]),
    "ucsa_composite_bwd_x2": (C.c_int32, [_p] * 17 + [_u32, _u32, _u32, _u32, _f] +
                              [_p] * 6),
    "ucsa_composite_bwd_f16": (C.c_int32, [_p] * 17 + [_u32, _u32, _u32, _u32, _f,
                                                        _f] + [_p] * 6),
    "ucsa_adam_step": (C.c_int32, [_p, _p, _p, _p, C.c_uint64, _u32, _f, _f,
                                   _f, _f, _f, _f, _p]),
    "ucsa_adam_step_scaled": (C.c_int32, [_p, _p, _p, _p, C.c_uint64, _u32,
                                          _f, _f, _f, _f, _f, _p, _p, _p, _p]),
    "ucsa_adam_count_skipped": (C.c_int32, [_p, _p, _p]),
    # ---- losses / post-processing / metric ----
    "ucsa_loss_partial_floats": (C.c_uint32, [_u32]),
    "ucsa_nerf_loss": (C.c_int32, [_p] * 6 + [_u32, _u32, _f, _f, _f, _f] +
                       [_p] * 6),
    "ucsa_nerf_loss_apply": (C.c_int32, [_p, _p, _p, _p, _p, _p, _u32, _u32, _p, _p, _p, _p, _f, _f, _p]),
    "ucsa_semantic_postproc": (C.c_int32, [_p, _u32, _u32, _p, _p, _p]),
    "ucsa_seg_tail": (C.c_int32, [_p, _p, _u32, _u32, _u32, _f, _p, _p, _p, _p,
                                  _p, _p]),
    "ucsa_confusion_matrix": (C.c_int32, [_p, _p, C.c_uint64, _u32, _p, _p]),
    "ucsa_bn_workspace_bytes": (C.c_uint64, [_u32, _u32]),
    "ucsa_bn_act_fwd": (C.c_int32, [_p, _p, _p, _p, _p, _p, _f, _f, _u32, _u32,
                                    C.c_int32, C.c_int32, C.c_int32, _p, _p, _p,
                                    _p, _p]),
    "ucsa_bn_act_bwd": (C.c_int32, [_p, _p, _p, _p, _p, _p, _u32, _u32, C.c_int32,
                                    C.c_int32, _p, _p, _p, _p, _p, _p]),
    # ---- occupancy-grid ray marching ----
    "ucsa_march_workspace_bytes": (C.c_uint64, [_u32]),
    "ucsa_march_rays_train": (C.c_int32, [_p, _p, _p, _f, _f, _f, _u32, _u32,
                                          _u32, _u32, _p, _p, _p, _p, _p, _p,
                                          _p, _u32, _p, _p]),
    "ucsa_composite_rays_train_fwd": (C.c_int32, [_p, _p, _p, _p, _p, _u32,
                                                  _u32, _u32, _p, _p, _p, _p,
                                                  _p]),
    "ucsa_composite_rays_train_bwd": (C.c_int32, [_p] * 9 + [_u32, _u32, _u32,
                                                            _p, _p, _p, _p]),
    "ucsa_march_rays": (C.c_int32, [_u32, _u32, _p, _p, _p, _p, _f, _f, _u32,
                                    _u32, _p, _f, _p, _p, _p, _p, _p, _u32,
                                    _p]),
    "ucsa_composite_rays": (C.c_int32, [_u32, _u32, _p, _p, _p, _p, _p, _p,
                                        _u32, _p, _p, _p, _p, _p]),
    "ucsa_compact_workspace_bytes": (C.c_uint64, [_u32]),
    "ucsa_compact_rays": (C.c_int32, [_u32, _p, _p, _p, _p, _p, _p, _p]),
    "ucsa_march_segment_workspace_bytes": (C.c_uint64, [_u32]),
    "ucsa_march_segment_stage_bytes": (C.c_uint64, [_u32, _u32]),
    "ucsa_march_segment_count": (C.c_int32, [_u32, _p, _u32, _p, _p, _p, _p,
                                             _f, _f, _u32, _u32, _p, _f, _p,
                                             _u32, _p, _p, _p, _p]),
    "ucsa_march_segment_write": (C.c_int32, [_u32, _p, _p, _p, _p, _p, _f, _f,
                                             _u32, _u32, _p, _f, _p, _u32, _p,
                                             _p, _p, _p, _p, _u32, _p]),
    "ucsa_march_segment_composite": (C.c_int32, [_u32, _p, _u32, _p, _p, _p,
                                                 _p, _f, _p, _p, _p, _u32, _p,
                                                 _p, _p, _p, _p]),
    "ucsa_march_segment_shade": (C.c_int32, [_u32, _p, _u32, _p, _p, _p, _p,
                                             _p, _f, _p, _p, _p, _p, _u32, _f,
                                             _p, _p, _p, _p, _p]),
    "ucsa_march_segment_shade_f16": (C.c_int32, [_u32, _p, _u32, _p, _p, _p,
                                                 _p, _p, _f, _p, _p, _p, _p,
                                                 _u32, _f, _p, _p, _p, _p,
                                                 _p]),
    "ucsa_march_segment_shade_h2": (C.c_int32, [_u32, _p, _u32, _p, _p, _p,
                                                _p, _p, _f, _p, _p, _p, _p,
                                                _u32, _f, _p, _p, _p, _p,
                                                _p]),
    "ucsa_march_segment_compact": (C.c_int32, [_u32, _p, _p, _p, _p, _p, _p,
                                               _p, _p]),
    "ucsa_march_train_fwd": (C.c_int32, [_p, _u32, _u32, _p, _p, _p, _f, _p,
                                         _p, _p, _p, _u32, _f, _p, _p, _p, _p,
                                         _p, _p, _p]),
    "ucsa_march_train_bwd": (C.c_int32, [_p, _u32, _u32, _p, _p, _p, _f, _p,
                                         _p, _p, _p, _p, _p, _p, _p, _u32, _f,
                                         _p, _p, _p, _p, _p, _p, _p, _p]),
    "ucsa_augment_workspace_bytes": (C.c_uint64, [_u32, _u32, _u32]),
    "ucsa_augment": (C.c_int32, [_p, _p, _u32, _u32, _u32,
                                 C.POINTER(AugParams), _u32, _u32, _p, _p, _p,
                                 _p]),
    "ucsa_density_grid_points": (C.c_int32, [_u32, _u32, _f, _u32, _p, _p]),
    "ucsa_density_grid_workspace_bytes": (C.c_uint64, []),
    "ucsa_density_grid_update": (C.c_int32, [_p, _p, C.c_uint64, _f, _f, _p,
                                             _p, _p]),
}

_lib: Optional[C.CDLL] = None


class UcsaError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libucsa_hip.so for gfx950 with hipcc (cross-compiles without a
    GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise UcsaError("building libucsa_hip.so failed:\n" +
                        (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UcsaError(
            f"{LIB_PATH} is missing.  This package has no CPU/PyTorch "
            "fallback: build the HIP library first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C {CSRC}`).")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise UcsaError(f"libucsa_hip.so does not export {name}; "
                            "rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ucsa_error_string(rc).decode()
        raise UcsaError(f"{what} failed: {msg} (code {rc})")


def make_grid(bound: float, n_levels: int = 16, log2_hashmap_size: int = 19,
              base_resolution: int = 16,
              per_level_scale: float = 2.0) -> Grid:
    g = Grid()
    check(lib().ucsa_grid_init(C.byref(g), bound, n_levels, log2_hashmap_size,
                               base_resolution, per_level_scale),
          "ucsa_grid_init")
    return g


def fvec(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])

"""Parameter holders standing in for the tiny-cuda-nn objects the reference
constructs (``tcnn.Encoding`` / ``tcnn.Network``,
nr4seg/nerf/network_tcnn_semantics.py:36-100).

Each is an ``nn.Module`` with one flat fp32 ``params`` Parameter in
tiny-cuda-nn's layout, so ``.parameters()`` feeds the two Adam groups the
reference builds (joint_train_lightning_net.py:899-915) and state_dict keys
are ``<name>.params``.  No arithmetic here: the HIP kernels consume ``params``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _lib


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


class HashGridEncoding(nn.Module):
    """tcnn.Encoding(3, {"otype": "HashGrid", ...}); params [entries*2]."""

    def __init__(self, bound: float, n_levels: int = 16,
                 n_features_per_level: int = 2, log2_hashmap_size: int = 19,
                 base_resolution: int = 16, per_level_scale: float = 2.0,
                 seed: int | None = None):
        super().__init__()
        if n_features_per_level != 2:
            raise ValueError("the HIP encoder implements n_features_per_level=2")
        self.n_input_dims = 3
        self.n_output_dims = n_levels * n_features_per_level
        self.grid = _lib.make_grid(bound, n_levels, log2_hashmap_size,
                                   base_resolution, per_level_scale)
        n = int(self.grid.total_entries) * 2
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        init = (torch.rand(n, generator=g) * 2.0 - 1.0) * 1e-4  # tcnn default
        self.params = nn.Parameter(init)

    def level_table(self):
        return [dict(scale=lv.scale, res=lv.res, entries=lv.entries,
                     offset=lv.offset, hashed=bool(lv.hashed))
                for lv in list(self.grid.level)[:self.grid.n_levels]]


class SHEncoding(nn.Module):
    """tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4}):
    no parameters; evaluated inside the composite kernel."""

    def __init__(self, degree: int = 4):
        super().__init__()
        if degree != 4:
            raise ValueError("the HIP path implements SH degree 4")
        self.n_input_dims = 3
        self.n_output_dims = degree * degree


class FullyFusedMLP(nn.Module):
    """tcnn.Network(n_in, n_out, FullyFusedMLP/ReLU/None, width, hidden).

    Bias-free; row-major [out,in] matrices back to back; input padded to a
    multiple of 16 with ones, output to a multiple of 16."""

    def __init__(self, n_input_dims: int, n_output_dims: int, n_neurons: int,
                 n_hidden_layers: int, kind: int,
                 gen: torch.Generator | None = None):
        super().__init__()
        if n_neurons != 64:
            raise ValueError("the HIP MLP kernels implement width 64")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.kind = kind
        self.shapes = [(n_neurons, _pad16(n_input_dims))]
        self.shapes += [(n_neurons, n_neurons)] * (n_hidden_layers - 1)
        self.shapes += [(_pad16(n_output_dims), n_neurons)]
        chunks = []
        for r, c in self.shapes:
            s = math.sqrt(6.0 / (r + c))  # xavier-uniform on padded shapes
            chunks.append((torch.rand(r * c, generator=gen) * 2.0 - 1.0) * s)
        self.params = nn.Parameter(torch.cat(chunks))

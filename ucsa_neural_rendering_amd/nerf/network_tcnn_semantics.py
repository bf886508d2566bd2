"""Mirror of reference ``nr4seg/nerf/network_tcnn_semantics.py``.

``SemanticNeRFNetwork`` keeps the reference constructor signature (:12-27),
attribute names (``encoder``, ``sigma_net``, ``encoder_dir``, ``color_net``,
``semantics_net``) and methods (``forward``, ``density``, ``color``,
``semantics``); the tiny-cuda-nn objects are replaced by parameter holders
whose arithmetic runs in the HIP kernels.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib, ops
from .encoding import FullyFusedMLP, HashGridEncoding, SHEncoding
from .renderer_semantics import SemanticNeRFRenderer


class SemanticNeRFNetwork(SemanticNeRFRenderer):

    def __init__(self, encoding="HashGrid", encoding_dir="SphericalHarmonics",
                 num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64,
                 num_layers_semantics=2, hidden_dim_semantics=64, bound=1,
                 num_semantic_classes=41, seed=None, **kwargs):
        super().__init__(bound, **kwargs,
                         num_semantic_classes=num_semantic_classes)
        if (num_layers, num_layers_color, num_layers_semantics,
                geo_feat_dim) != (2, 3, 2, 15):
            raise ValueError(
                "the HIP kernels implement the reference configuration: "
                "sigma 32->64->16, colour 32->64->64->3, semantics 16->64->C")
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        gen = torch.Generator().manual_seed(seed) if seed is not None else None

        # reference :34
        per_level_scale = float(np.exp2(np.log2(2048 * bound / 16) / (16 - 1)))
        self.encoder = HashGridEncoding(bound, 16, 2, 19, 16, per_level_scale,
                                        seed=seed)
        self.sigma_net = FullyFusedMLP(32, 1 + geo_feat_dim, hidden_dim,
                                       num_layers - 1, _lib.MLP_SIGMA, gen)
        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir = SHEncoding(4)
        self.in_dim_color = self.encoder_dir.n_output_dims + geo_feat_dim
        self.color_net = FullyFusedMLP(self.in_dim_color, 3, hidden_dim_color,
                                       num_layers_color - 1, _lib.MLP_COLOR,
                                       gen)
        self.num_layers_semantics = num_layers_semantics
        self.hidden_dim_semantics = hidden_dim_semantics
        self.in_dim_semantics = geo_feat_dim
        self.semantics_net = FullyFusedMLP(geo_feat_dim, num_semantic_classes,
                                           hidden_dim_semantics,
                                           num_layers_semantics - 1,
                                           _lib.MLP_SEM, gen)
        self._packed = {}

    # -- packed (MFMA A-fragment order) weights, refreshed when params change
    def _pack(self, name: str, net: FullyFusedMLP):
        p = net.params
        key = (p.data_ptr(), p._version)
        hit = self._packed.get(name)
        if hit is None or hit[0] != key or hit[1].device != p.device:
            out = None if hit is None or hit[1].device != p.device else hit[1]
            packed = ops.mlp_pack(net.kind, p, self.num_semantic_classes,
                                  out=out)
            self._packed[name] = (key, packed)
        return self._packed[name][1]

    def _field(self):
        return dict(grid=self.encoder.grid, table=self.encoder.params.detach(),
                    packed_sigma=self._pack("sigma", self.sigma_net),
                    packed_color=self._pack("color", self.color_net),
                    packed_sem=self._pack("sem", self.semantics_net))

    # -- pointwise API (reference :102-207) -----------------------------------
    def density(self, x):
        """x [M,3] in [-bound,bound] -> {"sigma" [M], "geo_feat" [M,15]}."""
        f = self._field()
        with torch.no_grad():
            feat = ops.hashgrid_encode_points(f["grid"], f["table"], x)
            h, sigma = ops.sigma_mlp_fwd(feat, f["packed_sigma"])
        return {"sigma": sigma, "geo_feat": h[:, 1:]}

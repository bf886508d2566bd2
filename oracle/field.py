"""Oracle (test infrastructure): fp32 CPU restatement of the Semantic-NeRF field.

Follows the reference module ``nr4seg/nerf/network_tcnn_semantics.py``:
constructor :12-100, ``forward`` :102-128, ``density`` :130-144, ``color``
:147-178, ``semantics`` :180-207, and ``nr4seg/nerf/activation.py:7-21``.

The three building blocks are tiny-cuda-nn objects in the reference
(HashGrid encoding, SphericalHarmonics encoding, FullyFusedMLP).  tiny-cuda-nn
is NOT under /root/reference (pip-installed from an unpinned git HEAD,
reference README.md:51), so their arithmetic is restated here from the
published algorithm (Mueller et al. 2022, "Instant Neural Graphics Primitives",
and the public tcnn documentation): PARITY UNPINNED, see oracle/__init__.py.

Conventions fixed by this restatement (and KAT-ed in tests/test_oracle_field.py):

* grid ``scale_l = 2^(l*log2(s)) * base - 1`` evaluated in float64 then cast to
  float32; ``res_l = ceil(scale_l) + 1``; level entries
  ``min(roundup8(res_l^3), 2^log2_hashmap_size)``;
* ``pos = x*scale_l + 0.5``; cell = floor(pos); trilinear weights from frac;
  corner c uses bit d of c for dimension d (bit 0 = x);
* dense index ``x + y*res + z*res^2`` when the level is not hashed, else
  ``x ^ y*2654435761 ^ z*805459861`` in uint32; then ``% entries``;
* features are level-major ``[l0f0, l0f1, l1f0, ...]``;
* MLPs have no biases, ReLU hidden, linear output; weight matrices are stored
  row-major ``[out, in]`` back to back in one flat vector; the input is padded
  to a multiple of 16 with the constant 1.0 and the output to a multiple of 16
  (extra rows sliced off);
* everything is fp32 (the reference's GPU path rounds to fp16; the HIP
  product has an fp32 mode checked tightly against this oracle and an fp16
  MFMA mode checked against it at a stated looser tolerance).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861


# --------------------------------------------------------------------------
# hash grid
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class GridLevel:
    scale: float  # float32 value
    res: int
    entries: int
    offset: int  # in entries
    hashed: bool


@dataclass(frozen=True)
class GridSpec:
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    levels: tuple

    @property
    def total_entries(self) -> int:
        last = self.levels[-1]
        return last.offset + last.entries

    @property
    def n_params(self) -> int:
        return self.total_entries * self.n_features

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * self.n_features


def make_grid_spec(bound: float = 4.0,
                   n_levels: int = 16,
                   n_features: int = 2,
                   log2_hashmap_size: int = 19,
                   base_resolution: int = 16,
                   per_level_scale: float | None = None) -> GridSpec:
    """Level table.  ``per_level_scale`` default follows reference
    network_tcnn_semantics.py:34: exp2(log2(2048*bound/16)/(16-1))."""
    if per_level_scale is None:
        per_level_scale = float(
            np.exp2(np.log2(2048 * bound / 16) / (16 - 1)))
    log2s = math.log2(per_level_scale)
    levels: List[GridLevel] = []
    offset = 0
    cap = 1 << log2_hashmap_size
    for l in range(n_levels):
        scale64 = math.pow(2.0, l * log2s) * base_resolution - 1.0
        # values that are integral up to float64 noise are snapped so the
        # resolution does not depend on libm rounding (SURVEY 8a caveat).
        if abs(scale64 - round(scale64)) < 1e-9 * max(1.0, abs(scale64)):
            scale64 = float(round(scale64))
        scale = float(np.float32(scale64))
        res = int(math.ceil(scale)) + 1
        dense_entries = res**3
        entries = (dense_entries + 7) // 8 * 8
        hashed = entries > cap
        entries = min(entries, cap)
        levels.append(GridLevel(scale, res, entries, offset, hashed))
        offset += entries
    return GridSpec(n_levels, n_features, log2_hashmap_size, base_resolution,
                    per_level_scale, tuple(levels))


def grid_index(spec: GridSpec, level: GridLevel, gx: torch.Tensor,
               gy: torch.Tensor, gz: torch.Tensor) -> torch.Tensor:
    """uint32 index arithmetic carried in int64 with explicit wrap."""
    M32 = 0xFFFFFFFF
    if level.hashed:
        idx = (gx & M32) ^ ((gy * PRIME_Y) & M32) ^ ((gz * PRIME_Z) & M32)
    else:
        idx = (gx + gy * level.res + gz * level.res * level.res) & M32
    return idx % level.entries


def hashgrid_encode(spec: GridSpec, x01: torch.Tensor,
                    params: torch.Tensor) -> torch.Tensor:
    """x01 [M,3] float32 in [0,1]; params flat [n_params] -> [M, L*F]."""
    assert x01.dtype == torch.float32
    table = params.view(-1, spec.n_features)
    outs = []
    for level in spec.levels:
        pos = x01 * level.scale + 0.5
        cell = torch.floor(pos)
        frac = pos - cell
        cell = cell.to(torch.int64)
        ws, idxs = [], []
        for c in range(8):
            w = torch.ones(x01.shape[0], dtype=torch.float32)
            g = []
            for d in range(3):
                if (c >> d) & 1:
                    w = w * frac[:, d]
                    g.append(cell[:, d] + 1)
                else:
                    w = w * (1.0 - frac[:, d])
                    g.append(cell[:, d])
            ws.append(w)
            idxs.append(grid_index(spec, level, g[0], g[1], g[2]) + level.offset)
        # ONE gather per level (autograd then builds one dense table gradient
        # per level instead of one per corner); the corners are still summed
        # in the order c = 0..7, so the values are unchanged bit for bit
        vals = table[torch.stack(idxs, dim=1)]                 # [M, 8, F]
        acc = torch.zeros(x01.shape[0], spec.n_features, dtype=torch.float32)
        for c in range(8):
            acc = acc + ws[c].unsqueeze(-1) * vals[:, c]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------
# spherical harmonics, degree 4 (16 outputs)
# --------------------------------------------------------------------------
def sh4_encode(d01: torch.Tensor) -> torch.Tensor:
    """d01 [M,3] in [0,1] (the reference maps (d+1)/2 first,
    network_tcnn_semantics.py:164); tcnn maps back with 2*d-1."""
    x = d01[:, 0] * 2.0 - 1.0
    y = d01[:, 1] * 2.0 - 1.0
    z = d01[:, 2] * 2.0 - 1.0
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = [
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ]
    return torch.stack(out, dim=-1)


# --------------------------------------------------------------------------
# bias-free fully-fused-style MLP
# --------------------------------------------------------------------------
def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


@dataclass(frozen=True)
class MLPSpec:
    n_in: int
    n_out: int
    width: int
    n_hidden: int  # number of hidden layers (>=1)

    @property
    def in_pad(self) -> int:
        return _pad16(self.n_in)

    @property
    def out_pad(self) -> int:
        return _pad16(self.n_out)

    @property
    def shapes(self):
        s = [(self.width, self.in_pad)]
        for _ in range(self.n_hidden - 1):
            s.append((self.width, self.width))
        s.append((self.out_pad, self.width))
        return s

    @property
    def n_params(self) -> int:
        return sum(a * b for a, b in self.shapes)


def mlp_split(spec: MLPSpec, params: torch.Tensor):
    mats = []
    o = 0
    for (r, c) in spec.shapes:
        mats.append(params[o:o + r * c].view(r, c))
        o += r * c
    return mats


def _q16(t: torch.Tensor) -> torch.Tensor:
    """Round to fp16 and back (emulates an fp16 MFMA operand).  Straight
    through for autograd: the gradient passes in fp32, unrounded.  (A plain
    ``t.half().float()`` would cast the GRADIENT to fp16 as well, WITHOUT a
    loss scale: at the tcnn-style initial state that flushed 70 % of the
    hash-grid gradient and all but 74 of the sigma net's 3072 entries to zero
    -- found by tests/scripts/tcnn_init_grad.py in round 4.)"""
    if not t.requires_grad:
        return t.half().float()
    return t + (t.half().float() - t).detach()


def mlp_forward(spec: MLPSpec, x: torch.Tensor, params: torch.Tensor,
                emulate_fp16: bool = False) -> torch.Tensor:
    """x [M, n_in] -> [M, n_out]; input padded with 1.0 to in_pad.
    emulate_fp16: weights and every layer's input rounded to fp16, products
    accumulated in fp32, outputs left in fp32 (the HIP fp16 option; tcnn's
    FullyFusedMLP numerics up to its fp16 output rounding)."""
    M = x.shape[0]
    if spec.in_pad != spec.n_in:
        ones = torch.ones(M, spec.in_pad - spec.n_in, dtype=x.dtype)
        x = torch.cat([x, ones], dim=-1)
    mats = mlp_split(spec, params)
    q = _q16 if emulate_fp16 else (lambda t: t)
    h = x
    for W in mats[:-1]:
        h = torch.relu(q(h) @ q(W).t())
    y = q(h) @ q(mats[-1]).t()
    return y[:, :spec.n_out]


def mlp_init(spec: MLPSpec, gen: torch.Generator) -> torch.Tensor:
    """Xavier-uniform per matrix on the padded shapes (tcnn default)."""
    chunks = []
    for (r, c) in spec.shapes:
        s = math.sqrt(6.0 / (r + c))
        chunks.append((torch.rand(r * c, generator=gen) * 2.0 - 1.0) * s)
    return torch.cat(chunks)


def grid_init(spec: GridSpec, gen: torch.Generator) -> torch.Tensor:
    """tcnn grid default: U(-1e-4, 1e-4)."""
    return (torch.rand(spec.n_params, generator=gen) * 2.0 - 1.0) * 1e-4


# --------------------------------------------------------------------------
# trunc_exp (reference nr4seg/nerf/activation.py:7-21)
# --------------------------------------------------------------------------
class _TruncExp(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


# --------------------------------------------------------------------------
# the field
# --------------------------------------------------------------------------
class OracleField:
    """Plain-tensor Semantic-NeRF field (reference SemanticNeRFNetwork minus
    the renderer base class).  Parameters are four flat fp32 tensors with
    ``requires_grad`` as set by the caller."""

    def __init__(self, bound: float = 4.0, num_semantic_classes: int = 40,
                 geo_feat_dim: int = 15, hidden_dim: int = 64,
                 seed: int | None = 123, grid_spec: GridSpec | None = None,
                 emulate_fp16: bool = False):
        self.emulate_fp16 = emulate_fp16
        # per-net emulation ("sigma", "color", "sem"): the training mode
        # train_precision="fp16" runs only the colour / semantics nets in fp16
        self.emulate_fp16_nets = ()
        self.bound = float(bound)
        self.C = num_semantic_classes
        self.geo_feat_dim = geo_feat_dim
        self.grid = grid_spec or make_grid_spec(bound)
        # sigma: L*F -> 1+geo, 1 hidden; colour: 16+geo -> 3, 2 hidden;
        # semantics: geo -> C, 1 hidden  (network_tcnn_semantics.py:48-100)
        self.sigma_spec = MLPSpec(self.grid.n_output_dims, 1 + geo_feat_dim,
                                  hidden_dim, 1)
        self.color_spec = MLPSpec(16 + geo_feat_dim, 3, hidden_dim, 2)
        self.sem_spec = MLPSpec(geo_feat_dim, num_semantic_classes,
                                hidden_dim, 1)
        if seed is not None:
            g = torch.Generator().manual_seed(seed)
            self.grid_params = grid_init(self.grid, g)
            self.sigma_params = mlp_init(self.sigma_spec, g)
            self.color_params = mlp_init(self.color_spec, g)
            self.sem_params = mlp_init(self.sem_spec, g)

    def _emu(self, name: str) -> bool:
        return bool(self.emulate_fp16) or name in getattr(self, "emulate_fp16_nets", ())

    def parameters(self):
        return [self.grid_params, self.sigma_params, self.color_params,
                self.sem_params]

    def requires_grad_(self, flag=True):
        for p in self.parameters():
            p.requires_grad_(flag)
        return self

    # reference density(): :130-144
    def density(self, x: torch.Tensor):
        x01 = (x + self.bound) / (2 * self.bound)
        if getattr(self, "fp16_table", False):
            # tiny-cuda-nn stores the table and the features in fp16 (fp32
            # interpolation in between); the casts pass gradients straight
            # through to the fp32 master parameters
            enc = _q16(hashgrid_encode(self.grid, x01, _q16(self.grid_params)))
        else:
            enc = hashgrid_encode(self.grid, x01, self.grid_params)
        h = mlp_forward(self.sigma_spec, enc, self.sigma_params,
                        self._emu("sigma"))
        sigma = trunc_exp(h[:, 0])
        return {"sigma": sigma, "geo_feat": h[:, 1:]}

    # reference color(): :147-178 (masked gather/scatter)
    def color(self, x, d, mask=None, geo_feat=None, **kw):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=torch.float32)
            if not mask.any():
                return rgbs
            d = d[mask]
            geo_feat = geo_feat[mask]
        d01 = (d + 1) / 2
        h = torch.cat([sh4_encode(d01), geo_feat], dim=-1)
        h = torch.sigmoid(mlp_forward(self.color_spec, h, self.color_params,
                                      self._emu("color")))
        if mask is not None:
            rgbs[mask] = h
            return rgbs
        return h

    # reference semantics(): :180-207 (softmax over masked rows only)
    def semantics(self, x, d, mask=None, geo_feat=None, **kw):
        if mask is not None:
            out = torch.zeros(mask.shape[0], self.C, dtype=torch.float32)
            if not mask.any():
                return out
            geo_feat = geo_feat[mask]
        h = mlp_forward(self.sem_spec, geo_feat, self.sem_params,
                        self._emu("sem"))
        p = torch.softmax(h, dim=-1)
        if mask is not None:
            out[mask] = p
            return out
        return p

    # reference forward(): :102-128
    def forward(self, x, d):
        den = self.density(x)
        return (den["sigma"], self.color(x, d, geo_feat=den["geo_feat"]),
                self.semantics(x, d, geo_feat=den["geo_feat"]))

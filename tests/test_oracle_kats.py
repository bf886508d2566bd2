"""Known-answer tests for the parts of the oracle that have no runnable
reference here: the tiny-cuda-nn restatement (hash grid / SH / MLP -- parity
unpinned, see oracle/__init__.py) and the CUDA slab test."""
import math

import numpy as np
import torch

from oracle import field as F
from oracle import losses as L
from oracle import rays as R

FLT_MAX = float(np.finfo(np.float32).max)


def test_grid_level_table_matches_survey():
    spec = F.make_grid_spec(bound=4.0)
    res = [lv.res for lv in spec.levels]
    assert res == [16, 25, 37, 56, 85, 128, 195, 295, 446, 676, 1024, 1553,
                   2353, 3566, 5405, 8192]
    ent = [lv.entries for lv in spec.levels]
    assert ent[:4] == [4096, 15632, 50656, 175616]
    assert all(e == 1 << 19 for e in ent[4:])
    assert [lv.hashed for lv in spec.levels] == [False] * 4 + [True] * 12
    assert spec.total_entries == 6537456
    assert spec.n_params == 13074912
    # integral scales are exact (levels 5, 10, 15)
    assert [spec.levels[i].scale for i in (0, 5, 10, 15)] == [15.0, 127.0,
                                                              1023.0, 8191.0]
    assert abs(spec.per_level_scale - 2**0.6) < 1e-12


def test_hash_and_dense_index():
    spec = F.make_grid_spec(bound=4.0)
    t = lambda v: torch.tensor([v], dtype=torch.int64)
    lv0 = spec.levels[0]
    assert int(F.grid_index(spec, lv0, t(3), t(5), t(7))) == 3 + 5 * 16 + 7 * 256
    # wrap: corner beyond the last cell on a dense level
    assert int(F.grid_index(spec, lv0, t(16), t(16), t(16))) == (16 + 256 + 4096) % 4096
    lv = spec.levels[8]
    x, y, z = 123, 45, 399
    want = (x ^ ((y * 2654435761) & 0xFFFFFFFF) ^ ((z * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    assert int(F.grid_index(spec, lv, t(x), t(y), t(z))) == want
    assert int(F.grid_index(spec, lv, t(1), t(0), t(0))) == 1
    assert int(F.grid_index(spec, lv, t(0), t(1), t(0))) == 2654435761 % (1 << 19)


def test_hashgrid_interpolates_linearly_and_reproduces_nodes():
    spec = F.make_grid_spec(bound=4.0, n_levels=2, log2_hashmap_size=19)
    g = torch.Generator().manual_seed(0)
    params = torch.randn(spec.n_params, generator=g)
    lv = spec.levels[0]  # scale 15: node k sits at x = (k - 0.5)/15
    node = torch.tensor([[(3 - 0.5) / 15, (4 - 0.5) / 15, (5 - 0.5) / 15]],
                        dtype=torch.float32)
    enc = F.hashgrid_encode(spec, node, params)
    idx = 3 + 4 * 16 + 5 * 256
    want = params.view(-1, 2)[idx]
    assert torch.allclose(enc[0, :2], want, atol=1e-5)
    # midpoint along x between nodes 3 and 4 = mean of the two node values
    mid = node.clone()
    mid[0, 0] = (3.5 - 0.5) / 15
    enc = F.hashgrid_encode(spec, mid, params)
    want = 0.5 * (params.view(-1, 2)[idx] + params.view(-1, 2)[idx + 1])
    assert torch.allclose(enc[0, :2], want, atol=1e-5)


def test_sh4_axis_values():
    d = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    sh = F.sh4_encode((d + 1) / 2)
    assert torch.allclose(sh[:, 0], torch.full((3,), 0.28209479177387814))
    # +z
    assert abs(float(sh[0, 2]) - 0.48860251190291987) < 1e-7
    assert abs(float(sh[0, 6]) - (0.94617469575755997 - 0.31539156525251999)) < 1e-7
    assert abs(float(sh[0, 12]) - 0.3731763325901154 * 2.0) < 1e-6
    # +x
    assert abs(float(sh[1, 3]) + 0.48860251190291987) < 1e-7
    assert abs(float(sh[1, 8]) - 0.54627421529603959) < 1e-7
    assert abs(float(sh[1, 15]) + 0.59004358992664352) < 1e-7
    # +y
    assert abs(float(sh[2, 1]) + 0.48860251190291987) < 1e-7
    assert abs(float(sh[2, 9]) - 0.59004358992664352) < 1e-7
    # orthonormality by Monte-Carlo quadrature over the sphere
    g = torch.Generator().manual_seed(1)
    v = torch.randn(200000, 3, generator=g, dtype=torch.float64)
    v = (v / v.norm(dim=-1, keepdim=True)).float()
    Y = F.sh4_encode((v + 1) / 2).double()
    gram = (Y.t() @ Y) / v.shape[0] * 4 * math.pi
    assert torch.allclose(gram, torch.eye(16, dtype=torch.float64), atol=0.03)


def test_mlp_padding_equivalence_and_counts():
    f = F.OracleField(seed=1)
    assert f.sigma_spec.n_params == 3072
    assert f.color_spec.n_params == 7168
    assert f.sem_spec.n_params == 4096
    spec = f.color_spec  # 31 -> pad 32 with constant one
    x = torch.randn(7, 31)
    y = F.mlp_forward(spec, x, f.color_params)
    W1, W2, W3 = F.mlp_split(spec, f.color_params)
    h = torch.relu(x @ W1[:, :31].t() + W1[:, 31])  # pad column acts as bias
    h = torch.relu(h @ W2.t())
    want = (h @ W3.t())[:, :3]
    assert torch.allclose(y, want, atol=1e-6)


def test_near_far_kats():
    aabb = torch.tensor([-4.0, -4, -4, 4, 4, 4])
    o = torch.tensor([[0.0, 0, 0],     # inside, +x
                      [0.0, 0, 0],     # inside, diagonal
                      [-10.0, 0, 0],   # outside, hits
                      [-10.0, 5, 0],   # outside, misses in y
                      [0.0, 0, 3.9],   # near clamp to min_near
                      [-10.0, 0, 5.0],  # misses in z
                      [1.0, 2.0, -3.0]])  # axis-parallel (1/0 = inf)
    d = torch.tensor([[1.0, 0, 0],
                      [0.6, 0.0, 0.8],
                      [1.0, 0, 0],
                      [1.0, 0, 0],
                      [0.0, 0, 1.0],
                      [1.0, 0, 0],
                      [0.0, 0.0, 1.0]])
    near, far = R.near_far_from_aabb(o, d, aabb)
    # ray 0: x-slab [-4,4]; y,z slabs are (-inf,inf) -> near=-4 -> clamp .2
    assert near[0] == np.float32(0.2) and far[0] == 4.0
    assert near[1] == np.float32(0.2) and abs(float(far[1]) - 5.0) < 1e-6
    assert near[2] == 6.0 and far[2] == 14.0
    assert near[3] == FLT_MAX and far[3] == FLT_MAX
    assert near[4] == np.float32(0.2) and abs(float(far[4]) - 0.1) < 1e-6
    assert near[5] == FLT_MAX and far[5] == FLT_MAX
    assert near[6] == np.float32(0.2) and far[6] == 7.0


def test_losses_match_the_torch_modules_the_reference_configures():
    """reference joint_train_lightning_net.py:37-45 builds these modules."""
    g = torch.Generator().manual_seed(3)
    B, N, C = 1, 50, 6
    rgb, gt = torch.rand(B, N, 3, generator=g), torch.rand(B, N, 3, generator=g)
    sem = torch.rand(B, N, C, generator=g)
    sem[0, :4] = 0  # invalid rows
    labels = torch.randint(0, C, (B, N), generator=g)
    depth = torch.rand(B, N, generator=g) * 3
    gtd = torch.rand(B, N, generator=g) * 3
    gtd[0, ::7] = 0
    lc, ls, ld = L.nerf_losses(rgb, sem, depth, gt, labels, gtd, 0.5)
    assert torch.allclose(lc, torch.nn.MSELoss(reduction="none")(rgb, gt).mean())
    s2 = sem.clone()
    inv = s2.sum(-1) == 0
    s2[inv] = 1
    s2 = s2 / s2.sum(-1, keepdim=True)
    lab = labels.clone()
    lab[inv] = -1
    want = torch.nn.NLLLoss(ignore_index=-1, reduction="none")(
        torch.log(s2 + 1e-15).permute(0, 2, 1), lab).mean()
    assert torch.allclose(ls, want)
    want = torch.nn.L1Loss(reduction="none")(depth[gtd != 0] / 0.5, gtd[gtd != 0]).mean(-1)
    assert torch.allclose(ld, want)
    # all-invalid branch
    _, ls0, _ = L.nerf_losses(rgb, sem * 0, depth, gt, labels, gtd, 0.5)
    assert ls0 is None
    logits = torch.randn(2, C, 5, 4, generator=g)
    lab = torch.randint(-1, C, (2, 5, 4), generator=g)
    loss, pred = L.seg_loss(logits, lab)
    want = torch.nn.CrossEntropyLoss(ignore_index=-1, reduction="none")(
        torch.softmax(logits, 1), lab).mean()
    assert torch.allclose(loss, want)


def test_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(100, generator=g)
    for wd in (0.0, 1e-6):
        p = p0.clone().requires_grad_()
        opt = torch.optim.Adam([p], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=wd)
        q, m, v = p0.clone(), torch.zeros(100), torch.zeros(100)
        for step in range(1, 4):
            grad = torch.randn(100, generator=g)
            p.grad = grad.clone()
            opt.step()
            q, m, v = L.adam_step(q, grad, m, v, step, 1e-2, weight_decay=wd)
            assert torch.allclose(q, p.detach(), rtol=1e-5, atol=1e-7)

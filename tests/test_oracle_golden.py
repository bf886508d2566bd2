"""Pin the CPU oracle against fixtures generated from the reference's own
Python (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import field as ofield
from oracle import metrics as ometrics
from oracle import rays as orays
from oracle import renderer as oren

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item())
            for k in z.files}


def test_trunc_exp_fwd_bwd():
    g = load("g0_trunc_exp.npz")
    x = g["x"].clone().requires_grad_()
    y = ofield.trunc_exp(x)
    y.backward(g["gy"])
    assert torch.equal(y.detach(), g["y"])
    assert torch.equal(x.grad, g["gx"])


def test_pose_permutation_and_rays():
    g = load("g1_rays.npz")
    for c2w, ref in zip(g["c2w"].numpy(), g["ngp_poses"].numpy()):
        assert np.array_equal(orays.nerf_matrix_to_ngp(c2w), ref)
    poses = g["ngp_poses"]
    o, d, n = orays.pixel_rays(poses, g["small_intr"].numpy(), 6, 8)
    assert torch.equal(o, g["small_o"])
    assert torch.allclose(d, g["small_d"], rtol=0, atol=1e-7)
    assert torch.equal(n, g["small_n"])
    o, d, n = orays.pixel_rays(poses[:1], g["big_intr"].numpy(), 480, 640)
    pick = g["big_pick"]
    assert torch.equal(o[:, pick], g["big_o"])
    assert torch.allclose(d[:, pick], g["big_d"], rtol=0, atol=1e-7)
    assert torch.equal(n[:, pick], g["big_n"])
    # training variant == full-image rays gathered at inds
    inds = torch.tensor([0, 5, 5, 47, 13])
    o2, d2, n2, ii = orays.pixel_rays_train(poses, g["small_intr"].numpy(), 6,
                                            8, inds)
    # (matmul blocking differs with the batch shape -> 1-ulp tolerance)
    assert torch.allclose(d2, orays.pixel_rays(poses, g["small_intr"].numpy(),
                                               6, 8)[1][:, inds], rtol=0,
                          atol=1e-7)
    assert torch.equal(n2, g["small_n"][:, inds])
    assert ii.shape == (3, 5)


def test_inverse_cdf_matches_sample_pdf():
    g = load("g3_sample_pdf.npz")
    z = oren.inverse_cdf(g["bins"], g["weights"], g["u"])
    assert torch.equal(z, g["samples"])


class _Table:
    """Same contract as make_golden.TableField (ids ride in geo_feat[:,0])."""

    def __init__(self, sig_c, sig_f, rgb_tab, prob_tab, C):
        self.sig = [sig_c, sig_f]
        self.rgb_tab, self.prob_tab, self.C = rgb_tab, prob_tab, C
        self.calls = 0
        self.base = [0, sig_c.numel()]

    def density(self, x):
        k = self.calls
        self.calls += 1
        s = self.sig[k].reshape(-1)
        geo = torch.zeros(s.numel(), 15)
        geo[:, 0] = torch.arange(s.numel(), dtype=torch.float32) + self.base[k]
        return {"sigma": s, "geo_feat": geo}

    def color(self, x, d, mask=None, geo_feat=None, **kw):
        ids = geo_feat[:, 0].long()
        out = torch.zeros(mask.shape[0], 3)
        out[mask] = self.rgb_tab[ids[mask]]
        return out

    def semantics(self, x, d, mask=None, geo_feat=None, **kw):
        ids = geo_feat[:, 0].long()
        out = torch.zeros(mask.shape[0], self.C)
        out[mask] = self.prob_tab[ids[mask]]
        return out


AABB = torch.tensor([-4.0, -4, -4, 4, 4, 4])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_run_table_field_forward_and_grads(tag):
    g = load(f"g4{tag}_run_table.npz")
    N, T, t, C = g["N"], g["T"], g["t"], g["C"]
    sig_c = g["sig_c"].clone().requires_grad_()
    sig_f = g["sig_f"].clone().requires_grad_()
    rgb = g["rgb_tab"].clone().requires_grad_()
    prob = g["prob_tab"].clone().requires_grad_()
    t_rand = g["t_rand"] if g["perturb"] else None
    u = g["u"] if t > 0 else None
    outs = {"image": [], "depth": [], "semantics": []}
    chunk = g["chunk"]
    for head in range(0, N, chunk):
        tail = min(head + chunk, N)
        ids = torch.cat([torch.arange(head * T, tail * T),
                         N * T + torch.arange(head * t, tail * t)])
        fld = _Table(sig_c[head:tail], sig_f[head:tail], rgb[ids], prob[ids],
                     C)
        r = oren.run(fld, g["rays_o"][None, head:tail],
                     g["rays_d"][None, head:tail],
                     g["norms"][None, head:tail], AABB, num_steps=T,
                     upsample_steps=t,
                     t_rand=None if t_rand is None else t_rand[head:tail],
                     u=None if u is None else u[head:tail])
        for k in outs:
            outs[k].append(r[k])
    res = {k: torch.cat(v, dim=1) for k, v in outs.items()}
    for k in ("image", "depth", "semantics"):
        assert torch.allclose(res[k], g[k], rtol=1e-6, atol=1e-7), k
    loss = (res["image"] * g["ci"]).sum() + (res["depth"] * g["cd"]).sum() + (
        res["semantics"] * g["cs"]).sum()
    loss.backward()
    assert torch.allclose(sig_c.grad, g["g_sig_c"], rtol=1e-5, atol=1e-7)
    if t > 0:
        assert torch.allclose(sig_f.grad, g["g_sig_f"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(rgb.grad, g["g_rgb"], rtol=1e-6, atol=1e-8)
    assert torch.allclose(prob.grad, g["g_prob"], rtol=1e-6, atol=1e-8)


@pytest.fixture(scope="module")
def golden_field():
    g = load("g5a_run_field.npz")
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=40, seed=123)
    gs = torch.Generator().manual_seed(int(g["grid_seed"]))
    fld.grid_params = (torch.rand(fld.grid.n_params, generator=gs) * 2 -
                       1) * float(g["grid_amp"])
    # guard against RNG drift between torch versions
    assert torch.equal(fld.grid_params[:8], g["grid_head"])
    assert torch.equal(fld.sigma_params[:8], g["sigma_head"])
    assert torch.equal(fld.color_params[:8], g["color_head"])
    assert torch.equal(fld.sem_params[:8], g["sem_head"])
    assert abs(float(fld.grid_params.double().sum()) -
               float(g["grid_sum"])) < 1e-6
    return fld


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_run_restated_field(tag, golden_field):
    g = load(f"g5{tag}_run_field.npz")
    T, t = g["T"], g["t"]
    with torch.no_grad():
        res = oren.render(golden_field, g["rays_o"][None], g["rays_d"][None],
                          g["norms"][None], AABB, staged=bool(g["staged"]),
                          max_ray_batch=g["chunk"],
                          t_rand=g["t_rand"][None] if g["perturb"] else None,
                          u=g["u"][None], num_steps=T, upsample_steps=t)
    for k in ("image", "depth", "semantics"):
        assert torch.allclose(res[k], g[k], rtol=1e-6, atol=1e-7), k


def test_semantics_meter():
    g = load("g7_meter.npz")
    C = g["C"]
    cm = ometrics.confusion(g["preds"].numpy(), g["truths"].numpy(), C)
    assert np.array_equal(cm, g["conf_mat"].numpy())
    miou, acc, cacc = ometrics.measure(cm)
    assert abs(miou - g["miou"]) < 1e-12
    assert abs(acc - g["total_acc"]) < 1e-12
    assert abs(cacc - g["class_avg_acc"]) < 1e-12


# ---- G6: NeRF losses, from the reference's forward_nerf_train itself ---------
def _g6_case(g, tag):
    B, C, H, W = 2, 3, *g[f"{tag}_img_fp16"].shape[-2:]
    bs, inds = int(g[f"{tag}_bs"]), g[f"{tag}_inds"]
    img = g[f"{tag}_img_fp16"][[bs]]
    gt_rgb = torch.gather(img.reshape(1, C, -1).permute(0, 2, 1), 1,
                          torch.stack(C * [inds], -1))
    labels = torch.gather(g[f"{tag}_seg"][[bs]].reshape(1, -1), 1, inds)
    gt_depth = torch.gather(g[f"{tag}_depth_fp16"][[bs]].reshape(1, -1), 1, inds)
    return gt_rgb, labels, gt_depth


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_nerf_losses_match_reference_forward_nerf_train(tag):
    """oracle.losses.nerf_losses / nerf_total_loss against values and gradients
    produced by the reference's own method (G6; case b: every ray invalid ->
    loss_semantics None)."""
    from oracle import losses as olosses
    from tests.util import maxabs
    g = load("g6_nerf_losses.npz")
    gt_rgb, labels, gt_depth = _g6_case(g, tag)
    image = g[f"{tag}_image"].clone().requires_grad_()
    depth = g[f"{tag}_depth"].clone().requires_grad_()
    sem = g[f"{tag}_sem"].clone().requires_grad_()
    lc, ls, ld = olosses.nerf_losses(image, sem, depth, gt_rgb, labels, gt_depth,
                                     float(g[f"{tag}_uom"]))
    assert (ls is None) == bool(g[f"{tag}_sem_is_none"])
    total = olosses.nerf_total_loss(lc, ls, ld)
    total.backward()
    assert abs(float(lc) - float(g[f"{tag}_loss_color"])) <= 1e-7
    assert abs(float(ld) - float(g[f"{tag}_loss_depth"])) <= 1e-6
    if ls is not None:
        assert abs(float(ls) - float(g[f"{tag}_loss_sem"])) <= 1e-5
    assert abs(float(total) - float(g[f"{tag}_total"])) <= 1e-6
    assert maxabs(image.grad, g[f"{tag}_g_image"]) <= 1e-9
    assert maxabs(depth.grad, g[f"{tag}_g_depth"]) <= 1e-9
    gs = torch.zeros_like(sem) if sem.grad is None else sem.grad
    assert maxabs(gs, g[f"{tag}_g_sem"]) <= 1e-7

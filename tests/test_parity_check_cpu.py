"""Test of the test: tests/parity_check.check_render (the causal rule of the
full-size GPU parity tests) on the CPU, with the oracle standing in for the
result under test.

* the oracle's own output passes (nothing is loose);
* an error of 3e-4 injected into 5 rays that sit on NO step fails;
* the same error injected into rays that DO have a candidate (a weight inside
  the mask window) fails as well -- having a candidate excuses nothing;
* a result in which a genuine threshold sample was decided the other way
  (its mask bit flipped: depth moves by > 2e-4 rel) passes, and is reported as
  explained by exactly that toggle;
* the same for a fine sample on sample_pdf's denom step placed by the other
  branch."""
import copy

import pytest
import torch

from oracle import renderer as oren
from tests import parity_check as pc
from tests.util import AABB4, lively_oracle_field, make_rays

N, T, t = 1024, 48, 48


@pytest.fixture(scope="module")
def case():
    fld = lively_oracle_field()
    o, d, n = make_rays(N, 5)
    u = torch.rand(N, t, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = oren.run(fld, o[None], d[None], n[None], AABB4, num_steps=T, upsample_steps=t,
                       u=u, return_aux=True)
    return fld, (o[None], d[None], n[None]), u, ref


def _res(ref):
    return {k: ref[k].clone() for k in ("image", "semantics", "depth")}


def test_the_oracle_itself_passes(case):
    fld, rays, u, ref = case
    out = pc.check_render(_res(ref), ref, fld, rays, AABB4, T, t, tag="self")
    assert out["loose"] == 0


@pytest.mark.parametrize("jitter", [False, True])
def test_an_injected_error_off_any_step_fails(case, jitter):
    """``jitter=True`` is how every GPU render test calls the checker:
    the moved-fine-sample alternative must not explain an injected error either."""
    fld, rays, u, ref = case
    am, ad = pc.flagged_a_priori(ref["aux"])
    clean = torch.nonzero(~(am | ad)).flatten()[:5]
    res = _res(ref)
    res["image"][0, clean, 1] += 3e-4
    with pytest.raises(AssertionError, match="NO alternative"):
        pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="inject-clean", jitter=jitter)


@pytest.mark.parametrize("jitter", [False, True])
def test_an_injected_error_on_a_flagged_ray_fails_too(case, jitter):
    fld, rays, u, ref = case
    am, _ = pc.flagged_a_priori(ref["aux"])
    flagged = torch.nonzero(am).flatten()[:5]
    assert len(flagged) == 5
    for key in ("semantics", "depth"):
        res = _res(ref)
        if key == "depth":
            res["depth"][0, flagged] *= 1.0 + 6e-4
        else:
            res["semantics"][0, flagged, 3] += 3e-4
        with pytest.raises(AssertionError, match="NO alternative"):
            pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="inject-flagged-" + key,
                            jitter=jitter)
    if jitter:      # ... nor one in the image or in the other semantics classes
        res = _res(ref)
        res["image"][0, flagged] += 3e-4
        with pytest.raises(AssertionError, match="NO alternative"):
            pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="inject-flagged-image", jitter=True)
        res = _res(ref)
        res["semantics"][0, flagged, 7] += 1.5e-4
        res["semantics"][0, flagged, 11] -= 1.2e-4
        with pytest.raises(AssertionError, match="NO alternative"):
            pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="inject-flagged-sem2", jitter=True)


def _toggle_case(case):
    """A ray + sample whose weight is inside the mask window and whose toggle
    moves the depth by more than the tolerance."""
    fld, rays, u, ref = case
    aux = ref["aux"]
    w, z = aux["weights"], aux["z"]
    near = (w - 1e-4).abs() <= pc.mask_window(aux)
    eff = w * z / rays[2][0].reshape(-1, 1) / ref["depth"][0][:, None].abs().clamp_min(1e-3)
    eff = torch.where(near, eff, torch.zeros_like(eff))
    i = int(eff.max(-1)[0].argmax())
    s = int(eff[i].argmax())
    assert float(eff[i, s]) > 2.5e-4, "fixture has no consequential threshold sample"
    return i, s


def test_a_genuine_threshold_flip_passes_and_is_named(case, capsys):
    fld, rays, u, ref = case
    i, s = _toggle_case(case)
    ro = pc.RayOracle(fld, rays[0][0, i], rays[1][0, i], rays[2][0, i], AABB4, T, t, u[i])
    with torch.no_grad():
        z, sigma, geo, xyz, _ = ro.sorted_samples()
        w, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
        mask = w > 1e-4
        mask[s] = ~mask[s]
        alt = ro.composite(z, w, rgbs, probs, mask)
    res = _res(ref)
    for k in res:
        res[k][0, i] = alt[k]
    out = pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="flip")
    assert out["loose"] == 1
    assert f"'mask_toggled': [{s}]" in capsys.readouterr().out
    # ... and the same ray with an extra 1e-4 on top of the flip does not
    res["image"][0, i, 0] += 1e-4
    for jitter in (False, True):
        with pytest.raises(AssertionError, match="NO alternative"):
            pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="flip+error", jitter=jitter)


def test_a_denom_step_sample_placed_by_the_other_branch_passes(case):
    """Fabricated: no fine sample of this small fixture sits within 1e-6 of the
    step (its emptiest bins have a cdf interval of 9e-5), so the window is
    widened for the test: the machinery -- move the sample, re-evaluate the
    density, re-sort, re-mask -- is what is exercised."""
    fld, rays, u, ref = case
    denom, _ = pc.fine_sample_cdf(ref["aux"])
    old = pc.DENOM_WINDOW
    try:
        pc.DENOM_WINDOW = 1e-4
        cand = torch.nonzero((denom - 1e-5).abs() <= pc.DENOM_WINDOW)
        if len(cand) == 0:
            pytest.skip("no fine sample near the denom step in this fixture")
        done = 0
        order = torch.argsort((denom - 1e-5).abs()[cand[:, 0], cand[:, 1]])
        for i, j in cand[order].tolist()[:6]:
            ro = pc.RayOracle(fld, rays[0][0, i], rays[1][0, i], rays[2][0, i], AABB4, T, t, u[i])
            with torch.no_grad():
                z, sigma, geo, xyz, _ = ro.sorted_samples((j,))
                w, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
                alt = ro.composite(z, w, rgbs, probs, w > 1e-4)
            res = _res(ref)
            for k in res:
                res[k][0, i] = alt[k]
            out = pc.check_render(res, ref, fld, rays, AABB4, T, t, tag=f"denom[{i},{j}]")
            done += out["loose"]
        assert done > 0, "no fabricated move was large enough to be loose"
        assert done > 0
    finally:
        pc.DENOM_WINDOW = old


def test_match_tolerance_follows_the_renders_ordinary_error_but_is_capped(case):
    """A render with an ordinary semantics error of ~3e-5 (a sharper field on
    the exact path does that): a genuine flip whose explained residual is
    2.5e-5 passes -- it looks like every other ray -- while a residual of 7e-5
    fails even though the noise level were higher still (cap: half of 1e-4)."""
    fld, rays, u, ref = case
    i, s = _toggle_case(case)
    ro = pc.RayOracle(fld, rays[0][0, i], rays[1][0, i], rays[2][0, i], AABB4, T, t, u[i])
    with torch.no_grad():
        z, sigma, geo, xyz, _ = ro.sorted_samples()
        w, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
        mask = w > 1e-4
        mask[s] = ~mask[s]
        alt = ro.composite(z, w, rgbs, probs, mask)
    g = torch.Generator().manual_seed(9)
    # heavy-tailed like the real thing: 3 % of the rays at +-3.2e-5, the rest at 1e-6
    noise = torch.where(torch.rand(N, generator=g) < 0.03, torch.full((N,), 3.2e-5),
                        torch.full((N,), 1e-6)) * torch.sign(torch.rand(N, generator=g) - 0.5)
    for extra, ok in ((2.5e-5, True), (7e-5, False)):
        res = _res(ref)
        res["semantics"][0, :, 3] += noise
        for k in res:
            res[k][0, i] = alt[k]
        res["semantics"][0, i, 3] += extra
        if ok:
            assert pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="noisy-flip")["loose"] == 1
        else:
            with pytest.raises(AssertionError, match="NO alternative"):
                pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="noisy-flip+error")
    # the same 2.5e-5 residual on a QUIET render exceeds ALT_ABS and fails
    res = _res(ref)
    for k in res:
        res[k][0, i] = alt[k]
    res["semantics"][0, i, 3] += 2.5e-5
    with pytest.raises(AssertionError, match="NO alternative"):
        pc.check_render(res, ref, fld, rays, AABB4, T, t, tag="quiet-flip+2.5e-5")


def test_a_fine_sample_moved_within_its_round_off_is_named_and_a_larger_move_is_not(case):
    """The third alternative (round 5, `jitter=True`): a result in which ONE fine
    sample sits 4 x its modelled depth round-off away (field re-evaluated there) is
    reproduced by moving that sample, ~4 units; the same at 20 units (beyond
    JITTER = 6) is not; and by default neither is.  Match tolerances scaled down to
    a quarter of the move's effect: the small fixture has no ray on which 4 units move
    an output by 1e-4."""
    fld, rays, u, ref = case
    tried = 0
    for i in range(0, N, 37):
        ro = pc.RayOracle(fld, rays[0][0, i], rays[1][0, i], rays[2][0, i], AABB4, T, t, u[i])
        with torch.no_grad():
            dz = pc.fine_depth_noise(ro)
            z0, sigma0, geo0, xyz0, order0 = ro.sorted_samples()
            w0, rgbs0, probs0 = ro.shade_all(z0, sigma0, geo0, xyz0)
            base = ro.composite(z0, w0, rgbs0, probs0, w0 > 1e-4)
            rank_of = torch.empty(T + t, dtype=torch.long)
            rank_of[order0[0]] = torch.arange(T + t)
            j = int((dz * w0[rank_of[T:]]).argmax())

            def moved(units):
                z, sigma, geo, xyz, _ = ro.sorted_samples(shift={j: units * float(dz[j])})
                w, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
                return ro.composite(z, w, rgbs, probs, w > 1e-4)

            got4, got20 = moved(4.0), moved(20.0)
            e4, e20 = pc._errors(got4, base), pc._errors(got20, base)
            if max(e4[:2]) < 1.5e-6 or max(e20) < 3 * max(e4):
                continue            # nothing to see on this ray / the response saturates
            tried += 1
            # (a third of the effect, and not below the fp32 noise of the composite itself)
            tol = (max(e4[0] / 3, 4e-7), max(e4[1] / 3, 4e-7), max(e4[2] / 3, 2e-6))
            score, what, errs, _ = pc.explain_ray(ro, got4, tol=tol, jitter=True)
            assert score <= 1.0 and what["moved_fine_sample"]["sample"] == j, (i, score, what, e4)
            assert 2.5 < what["moved_fine_sample"]["in_dz"] < 5.5, what
            score, what, errs, _ = pc.explain_ray(ro, got20, tol=tol, jitter=True)
            assert score > 1.0, (i, score, what)
            score, what, errs, _ = pc.explain_ray(ro, got4, tol=tol)
            assert score > 1.0 and "moved_fine_sample" not in what
        if tried == 3:
            break
    assert tried == 3


def test_the_oracle_with_its_cdf_summed_in_another_order_is_fully_explained():
    """Real round-off instead of an injected error: the oracle against itself,
    the second render with sample_pdf's cdf as a sequential fp32 running sum
    (torch.cumsum accumulates in double on the CPU; a GPU's parallel scan rounds
    differently again).  4096 rays: a few end up above the stated tolerance -- a
    mask flip, and fine samples in nearly empty bins that sit up to 2e-3 elsewhere --
    and every one of them must be matched by a named alternative (the moved-sample
    one included: what the whole-view GPU test runs with)."""
    n_rays, T2, t2 = 4096, 48, 48
    fld = lively_oracle_field()
    o, d, n = make_rays(n_rays, 5)
    u = torch.rand(n_rays, t2, generator=torch.Generator().manual_seed(1))
    rays = (o[None], d[None], n[None])

    def seq_cdf(bins, weights, uu):
        w = weights + 1e-5
        pdf = w / torch.sum(w, -1, keepdim=True)
        acc, cols = torch.zeros_like(pdf[:, 0]), []
        for k in range(pdf.shape[1]):
            acc = acc + pdf[:, k]
            cols.append(acc)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.stack(cols, -1)], -1)
        hi = torch.searchsorted(cdf, uu.contiguous(), right=True)
        lo = torch.clamp(hi - 1, min=0)
        hi = torch.clamp(hi, max=cdf.shape[-1] - 1)
        c0, c1 = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
        b0, b1 = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
        denom = c1 - c0
        denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
        return b0 + (uu - c0) / denom * (b1 - b0)

    orig = oren.inverse_cdf
    with torch.no_grad():
        ref = oren.run(fld, *rays, AABB4, num_steps=T2, upsample_steps=t2, u=u, return_aux=True)
        try:
            oren.inverse_cdf = seq_cdf
            alt = oren.run(fld, *rays, AABB4, num_steps=T2, upsample_steps=t2, u=u)
        finally:
            oren.inverse_cdf = orig
    moved = (alt["depth"] != ref["depth"]).float().mean()
    assert float(moved) > 0.2, "the two summation orders gave the same render"
    out = pc.check_render(_res(alt), ref, fld, rays, AABB4, T2, t2, tag="seq-cdf", jitter=True)
    assert out["loose"] >= 1 and out["by_jitter"] >= 1, out
    # ... the rays it explains are counted against a bound (here: none allowed)
    with pytest.raises(AssertionError, match="moved-fine-sample"):
        pc.check_render(_res(alt), ref, fld, rays, AABB4, T2, t2, tag="seq-cdf-bound", jitter=True,
                        max_by_jitter=0)
    # ... and without the moved-sample alternative the same render does NOT pass
    with pytest.raises(AssertionError, match="NO alternative"):
        pc.check_render(_res(alt), ref, fld, rays, AABB4, T2, t2, tag="seq-cdf-no-moved-sample")

"""Full-size render parity: the stated tolerance, and a CAUSAL explanation for
every ray above it (test infrastructure; used by tests/test_gpu_configs.py,
tested on the CPU by tests/test_parity_check_cpu.py).

Stated tolerance (fp32): image / semantics 1e-4 abs, depth 2e-4 rel -- for at
least 99.5 % of the rays; every ray within 2e-3 / 5e-3; median <= 5e-6.  What is
ENFORCED for the 99.5 %: a hard 0.5 % for a check of >= 32 768 rays (and for the whole
307 200-ray view); for a smaller SAMPLE of a view, 0.5 % + three binomial standard
deviations of the sample count (4096 rays: 34 = 0.83 %; never below 8).

Why not 100 % at thousands of rays: the reference has two STEP functions in
this path, and a ray that sits on one is decided by fp32 round-off:

1. the mask ``weights > 1e-4`` (reference renderer_semantics.py:249-250): a
   sample at the threshold is shaded on one side and dropped on the other;
2. ``sample_pdf``'s ``denom < 1e-5 -> 1`` (:40-41): an empty bin's pdf is
   1e-5 / (sum(w) + (T-2) 1e-5), within 1e-8 of that threshold on an opaque
   ray, so a fine sample landing in one is placed by either branch.

Round 3 excused a loose ray when it HAD such a sample (true for 23 % of all
rays before looking at the result).  Now the explanation is causal: for every
loose ray the ORACLE is re-evaluated on that ray with the candidate decisions
taken the other way -- each subset of the (<= 3) samples whose weight is
within the fp32 noise of 1e-4 toggled in the mask, each subset of the (<= 2)
fine samples whose cdf interval is within 1e-6 of 1e-5 placed by the other
branch (which moves the sample, so density, sort, weights and masks are
recomputed) -- and the result under test must MATCH ONE ALTERNATIVE in all
three outputs at once: image and semantics within ``ALT_ABS``, depth within
``ALT_REL``.  There is no blanket escape: a loose ray that matches no
alternative fails the test, wherever it sits.

The match tolerance is ``ALT_ABS`` / ``ALT_REL`` or, where the render's ORDINARY
error is larger than that, TWICE the 99.5th percentile of the error of its rays
that are NOT loose -- never more than half the stated tolerance (``alt_tolerances``):
once its flipped decision is accounted for, a loose ray must look like every
other ray of the same render.  (A field trained to sharper class logits has an
ordinary semantics error of 2.5e-5 ... 3e-5 at that percentile on the exact f32
path; a fixed 2e-5 then rejected rays whose explained residual was 2.2e-5 / 2.8e-5.)

"Within the fp32 noise of 1e-4" is ``mask_window``: 6 x the modelled round-off of
the DEPTHS, at least ``window_floor`` = four times the render's median image error (the
nets' arithmetic moves the weights too; round 5), at most 2 % of the threshold.

A third alternative, used by the GPU render tests (``jitter=True``): ONE fine
sample moved by at most 6 x the modelled round-off of its depth, the field
re-evaluated there (``_moved_fine_sample``) -- a one-parameter family, not a
decision; it explains the 0-2 semantics-only rays of a 307 200-ray view at 96+96
samples, and most of the (few) loose rays of a render at 16+16 samples, where a bin
is 0.4 wide; ``check_render`` reports how many rays it explained (``by_jitter``).
"""
from __future__ import annotations

import itertools
import math

import torch

from oracle import renderer as oren
from oracle.rays import near_far_from_aabb

TOL_ABS, TOL_DEPTH_REL = 1e-4, 2e-4          # the stated tolerance
CAP_ABS, CAP_DEPTH_REL = 2e-3, 5e-3          # no ray beyond
ALT_ABS, ALT_REL = 2e-5, 5e-5                # match to an alternative
MAX_MASK_CAND, MAX_DENOM_CAND = 3, 2
DENOM_WINDOW = 1e-6


def fine_sample_cdf(aux):
    """``inverse_cdf`` recomputed from the oracle's own bins / weights / u:
    per fine sample the cdf interval ``denom = c1 - c0`` and the bin width."""
    w = aux["w_coarse"][:, 1:-1] + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = aux["u"].contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(hi - 1, min=0)
    hi = torch.clamp(hi, max=cdf.shape[-1] - 1)
    denom = torch.gather(cdf, 1, hi) - torch.gather(cdf, 1, lo)
    bins = aux["z_mid_coarse"]
    width = torch.gather(bins, 1, hi) - torch.gather(bins, 1, lo)
    return denom, width


def depth_noise(aux):
    """Per sorted sample: how far fp32 round-off can move its DEPTH.  A coarse
    depth carries ~2 ulp(z).  A FINE depth is ``b0 + (u - c0) / denom * (b1 -
    b0)``: the cdf is an fp32 running sum of ~T terms (round-off of a few 2^-23,
    different between a sequential and a parallel scan), so the depth moves by
    ``4 * 2^-23 / denom * (b1 - b0)`` on top."""
    z = aux["z"]
    ulp = torch.exp2(torch.floor(torch.log2(z.abs().clamp_min(1e-30))) - 23)
    dz = 2 * ulp
    if "w_coarse" in aux:
        denom, width = fine_sample_cdf(aux)
        denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
        dz_fine = 4 * 2.0 ** -23 / denom * width.abs()
        T = aux["w_coarse"].shape[1]
        dz_cat = torch.cat([torch.zeros(z.shape[0], T), dz_fine], -1)
        dz = dz + torch.gather(dz_cat, 1, aux["order"])
    return dz


def weight_noise(aux):
    """Per sorted sample: how far fp32 round-off can move its weight.

    ``alpha_s ~ sigma_s * (z[s+1] - z[s])``, so ``dw/w ~ (dz_s + dz_s+1) /
    delta_s`` with ``depth_noise``'s dz.  Plus the
    transmittance in front of the sample: ``dT/T = sum_{j<s} x_j d(delta_j) /
    delta_j`` with ``x_j = sigma_j delta_j`` -- the interval noise of every
    sample in front, amplified by its optical depth.  Plus the cancellation in
    ``alpha`` itself (below).  Returned: the
    un-clamped estimate of |dw| per sorted sample."""
    z, w = aux["z"], aux["weights"]
    dz = depth_noise(aux)
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    dz_next = torch.cat([dz[:, 1:], torch.zeros_like(dz[:, :1])], -1)
    r = ((dz + dz_next) / delta.clamp_min(1e-12)).double()
    w64 = w.double()
    T_front = (1.0 - (torch.cumsum(w64, -1) - w64)).clamp_min(1e-12)
    alpha = (w64 / T_front).clamp(0.0, 1.0 - 1e-12)
    x = -torch.log1p(-alpha)
    up = torch.cumsum(x * r, -1) - x * r
    # alpha = 1 - exp(-x): the subtrahend exp(-x) ~ 1 is rounded to half an ulp(1)
    # = 6e-8 on each side (more with a fast exp), an ABSOLUTE error of alpha -- 0.1 %
    # of a weight of 1e-4 behind empty space, where the terms above are ~0 (round 5)
    d_alpha = 2.0 ** -23
    return (w64 * (r + up) + T_front * d_alpha).to(w.dtype)


WINDOW_MIN, WINDOW_MAX = 1e-7, 2e-6


def mask_window(aux, floor=WINDOW_MIN):
    """Per sorted sample: the half-width around 1e-4 inside which round-off
    decides the mask -- 6 x the modelled noise (its constants are estimates of
    an order-dependent round-off), at least ``floor`` (>= 1e-7), at most 2 % of
    the threshold (largest distance of a sample observed flipped on the GPU: 1.2 %)."""
    return (6.0 * weight_noise(aux)).clamp(min(max(floor, WINDOW_MIN), WINDOW_MAX), WINDOW_MAX)


def window_floor(e_img, loose):
    """The modelled noise above is the DEPTHS' round-off only.  The nets'
    arithmetic (bf16x3 / f16x2 MFMA against the oracle's fp32 BLAS) moves every
    weight too -- directly, and through the coarse pass's pdf, which places the fine
    samples -- and the outputs are sums of weight x O(1): a weight is not known
    better than the render's outputs are.  Floor of the window = FOUR times the
    median image error over the ordinary rays (a view has 6e7 samples: the flips
    that matter are the 3.5-sigma tail of the weight error, and the median output
    error is about one sigma of it), inside [1e-7, 2e-6] like the window itself:
    4.5e-7 on the f16x2 whole view -> 1.8e-6; 1e-7 on the exact path -> 4e-7.
    (Round 5: on fields that leave haze in empty space the whole-view test met up to
    7 rays with a semantics error of 1.00e-4 ... 1.02e-4 -- one sample of class
    probability 1.0 flipped -- and NO candidate in the depth-only window; with a
    floor of twice the median, still 4 flips 1.4-1.6 % from the threshold.  The same
    picture without a GPU: tests/scripts/oracle_self_noise.py.)"""
    ok = ~loose
    if int(ok.sum()) < 16:
        return WINDOW_MIN
    return min(max(4.0 * float(e_img[ok].median()), WINDOW_MIN), WINDOW_MAX)


def _inverse_cdf_flipped(bins, weights, u, flip):
    """oracle.renderer.inverse_cdf with the ``denom < 1e-5`` decision of the
    fine samples in ``flip`` [N,t] taken the other way."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(hi - 1, min=0)
    hi = torch.clamp(hi, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b0, b1 = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    denom = c1 - c0
    small = (denom < 1e-5) ^ flip
    denom = torch.where(small, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


class RayOracle:
    """The oracle's ``run`` on ONE ray, split at the two step functions so that
    either can be decided by hand (follows oracle/renderer.run line by line)."""

    def __init__(self, fld, o, d, nrm, aabb, T, t, u_row, t_rand_row=None,
                 density_scale=1.0, min_near=0.2):
        self.fld, self.aabb, self.T, self.t = fld, aabb, T, t
        self.o, self.d, self.nrm = o.view(1, 3), d.view(1, 3), nrm.view(1)
        self.ds = density_scale
        nears, fars = near_far_from_aabb(self.o, self.d, aabb, min_near)
        self.zc = oren.coarse_z(nears.unsqueeze(-1), fars.unsqueeze(-1), T,
                                None if t_rand_row is None else t_rand_row.view(1, T))
        self.xyz_c = oren.positions(self.o, self.d, self.zc, aabb)
        den = fld.density(self.xyz_c.reshape(-1, 3))
        self.sig_c = den["sigma"].view(1, T)
        self.geo_c = den["geo_feat"].view(1, T, -1)
        self.u = None if u_row is None else u_row.view(1, t)
        if t > 0:
            deltas, w = oren.alpha_weights(self.zc, self.sig_c, self.ds)
            self.z_mid = self.zc[:, :-1] + 0.5 * deltas[:, :-1]
            self.w_c = w

    def sorted_samples(self, denom_flip=None, shift=None):
        """(z, sigma, geo, xyz) of the merged, sorted samples; ``denom_flip``:
        indices of fine samples placed by the other branch of the denom step;
        ``shift`` {fine sample: eps}: those fine samples moved by eps along the ray
        (the field is re-evaluated there)."""
        if self.t == 0:
            return self.zc, self.sig_c, self.geo_c, self.xyz_c, None
        flip = torch.zeros(1, self.t, dtype=torch.bool)
        for j in (denom_flip or ()):
            flip[0, j] = True
        new_z = _inverse_cdf_flipped(self.z_mid, self.w_c[:, 1:-1], self.u, flip)
        for j, eps in (shift or {}).items():
            new_z[0, j] = (new_z[0, j].double() + eps).float()
        new_xyz = oren.positions(self.o, self.d, new_z, self.aabb)
        den2 = self.fld.density(new_xyz.reshape(-1, 3))
        z = torch.cat([self.zc, new_z], 1)
        z, order = torch.sort(z, dim=1)
        xyz = torch.gather(torch.cat([self.xyz_c, new_xyz], 1), 1,
                           order.unsqueeze(-1).expand(-1, -1, 3))
        sigma = torch.gather(torch.cat([self.sig_c, den2["sigma"].view(1, self.t)], 1), 1, order)
        geo_all = torch.cat([self.geo_c, den2["geo_feat"].view(1, self.t, -1)], 1)
        geo = torch.gather(geo_all, 1, order.unsqueeze(-1).expand_as(geo_all))
        return z, sigma, geo, xyz, order

    def shade_all(self, z, sigma, geo, xyz):
        """weights and the nets' outputs on EVERY sample (no mask)."""
        _, weights = oren.alpha_weights(z, sigma, self.ds)
        S = z.shape[1]
        dirs = self.d.view(1, 1, 3).expand(1, S, 3).reshape(-1, 3)
        every = torch.ones(S, dtype=torch.bool)
        fg = geo.reshape(S, -1)
        rgbs = self.fld.color(xyz.reshape(-1, 3), dirs, mask=every, geo_feat=fg).view(S, 3)
        probs = self.fld.semantics(xyz.reshape(-1, 3), dirs, mask=every, geo_feat=fg).view(S, -1)
        return weights[0], rgbs, probs

    def composite(self, z, weights, rgbs, probs, mask):
        w = torch.where(mask, weights, torch.zeros_like(weights))
        return {"depth": (w * z[0]).sum() / self.nrm[0],
                "image": (w[:, None] * rgbs).sum(0),
                "semantics": (w[:, None] * probs).sum(0)}


def _errors(got, ref):
    return (float((got["image"].double() - ref["image"].double()).abs().max()),
            float((got["semantics"].double() - ref["semantics"].double()).abs().max()),
            float((got["depth"].double() - ref["depth"].double()).abs() /
                  ref["depth"].double().abs().clamp_min(1e-3)))


def alt_tolerances(e_img, e_sem, rel, loose):
    """(image, semantics, depth) match tolerances of a render: twice its ordinary
    error level (p99.5 over the rays that are not loose), at least ALT_ABS /
    ALT_REL, at most half the stated tolerance."""
    ok = ~loose
    if int(ok.sum()) < 16:
        return ALT_ABS, ALT_ABS, ALT_REL
    # twice the percentile: the rays in question ARE the tail of the distribution
    q = lambda e, lo, hi: min(max(2.0 * float(e[ok].quantile(0.995)), lo), hi)  # noqa: E731
    return (q(e_img, ALT_ABS, 0.5 * TOL_ABS), q(e_sem, ALT_ABS, 0.5 * TOL_ABS),
            q(rel, ALT_REL, 0.5 * TOL_DEPTH_REL))


def explain_ray(ro: RayOracle, got, window_fn=None, tol=(ALT_ABS, ALT_ABS, ALT_REL),
                floor=WINDOW_MIN, jitter=False):
    """Best alternative for one ray: (score, description, errors).  score <= 1
    means ``got`` matches that alternative within ``tol`` (image, semantics,
    depth; default ALT_ABS / ALT_REL) in all three outputs.  Alternatives: subsets of the denom-step fine samples
    flipped x subsets of the at-threshold samples toggled."""
    with torch.no_grad():
        denom_cand = []
        if ro.t > 0:
            aux1 = {"w_coarse": ro.w_c, "z_mid_coarse": ro.z_mid, "u": ro.u}
            denom, _ = fine_sample_cdf(aux1)
            near = (denom[0] - 1e-5).abs()
            idx = torch.nonzero(near <= DENOM_WINDOW).flatten()
            idx = idx[torch.argsort(near[idx])][:MAX_DENOM_CAND]
            denom_cand = idx.tolist()
        best = (float("inf"), None, None)
        n_alt = 0
        nearest = []
        for kd in range(len(denom_cand) + 1):
            for dflip in itertools.combinations(denom_cand, kd):
                z, sigma, geo, xyz, order = ro.sorted_samples(dflip)
                weights, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
                aux = {"z": z, "weights": weights[None]}
                if ro.t > 0:     # the window needs the merged order of THIS alternative
                    aux.update(w_coarse=ro.w_c, z_mid_coarse=ro.z_mid, u=ro.u, order=order)
                win = (window_fn(aux) if window_fn else mask_window(aux, floor))[0]
                dist = (weights - 1e-4).abs()
                if not dflip:    # (diagnostics: the two weights nearest the threshold, their windows)
                    near2 = torch.argsort(dist)[:2].tolist()
                    nearest = [(float(weights[s]), float(win[s])) for s in near2]
                cand = torch.nonzero(dist <= win).flatten()
                cand = cand[torch.argsort((dist / win)[cand])][:MAX_MASK_CAND].tolist()
                base = weights > 1e-4
                for km in range(len(cand) + 1):
                    for tog in itertools.combinations(cand, km):
                        mask = base.clone()
                        for s in tog:
                            mask[s] = ~mask[s]
                        alt = ro.composite(z, weights, rgbs, probs, mask)
                        ei, es, ed = _errors(got, alt)
                        score = max(ei / tol[0], es / tol[1], ed / tol[2])
                        n_alt += 1
                        if score < best[0]:
                            best = (score, {"denom_flipped": list(dflip), "mask_toggled": list(tog),
                                            "toggled_weights": [float(weights[s]) for s in tog]},
                                    (ei, es, ed))
        if best[0] > 1.0 and jitter and ro.t > 0:
            for score, j, eps, units, errs in _moved_fine_sample(ro, got, tol):
                n_alt += 1
                if score < best[0]:
                    best = (score, {"denom_flipped": [], "mask_toggled": [], "toggled_weights": [],
                                    "moved_fine_sample": {"sample": j, "eps": eps, "in_dz": round(units, 2)}},
                            errs)
        best[1]["nearest_w_win"] = [(round(a * 1e4, 4), round(b * 1e4, 4)) for a, b in nearest]
        return best + (n_alt,)


JITTER = 6.0        # the same multiple of the modelled round-off as the mask window


def fine_depth_noise(ro):
    """Per fine sample of one ray: ``depth_noise``'s round-off of its depth -- 2 ulp(z)
    plus ``4 * 2^-23 / denom * (b1 - b0)``: a sample drawn into a nearly EMPTY bin
    (pdf = 1e-5 / sum, cdf interval ~2e-5 on a ray that is not opaque) is placed by
    the ratio of two differences of cdf values near 0.5, and 4 ulp of the cdf's
    running sum move it by ~2 % of the bin: 1e-3 at a bin width of 0.06."""
    aux1 = {"w_coarse": ro.w_c, "z_mid_coarse": ro.z_mid, "u": ro.u}
    denom, width = fine_sample_cdf(aux1)
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    z = _inverse_cdf_flipped(ro.z_mid, ro.w_c[:, 1:-1], ro.u, torch.zeros(1, ro.t, dtype=torch.bool))
    ulp = torch.exp2(torch.floor(torch.log2(z.abs().clamp_min(1e-30))) - 23)
    return (2 * ulp + 4 * 2.0 ** -23 / denom * width.abs())[0]


def _moved_fine_sample(ro, got, tol, top=4):
    """Third named cause (round 5), after the two step functions: ONE fine sample
    sits elsewhere in its bin, by at most ``JITTER`` x the modelled round-off of its
    depth (``fine_depth_noise``), and the FIELD is re-evaluated there -- density,
    sort, weights, masks, colour, class probabilities.  Such a sample has moved by up
    to 1e-3 between a sequential and a parallel running sum of the same pdf (measured:
    the oracle against itself with the cdf summed in the other order); where it
    carries a weight of 1e-2 next to a class boundary (probability 0.87, logits
    steep), the semantics move by 1e-4 while colour and depth barely do.  Not a
    decision but a one-parameter family, bounded by the fp32 round-off of the
    reference's own formula, that must reproduce all three outputs at once.  Yields
    (score, fine sample, eps, eps / dz, errors) for the ``top`` samples by weight x
    round-off, the step from a linear estimate refined once, verified exactly."""
    dz = fine_depth_noise(ro).double()
    z0, sigma0, geo0, xyz0, order0 = ro.sorted_samples()
    w0, _, _ = ro.shade_all(z0, sigma0, geo0, xyz0)
    rank_of = torch.empty(ro.T + ro.t, dtype=torch.long)
    rank_of[order0[0]] = torch.arange(ro.T + ro.t)
    impact = dz * w0[rank_of[ro.T:]].double().clamp_min(1e-6)

    def outputs(shift):
        z, sigma, geo, xyz, _ = ro.sorted_samples(shift=shift)
        w, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
        return ro.composite(z, w, rgbs, probs, w > 1e-4)

    def vec(a, b):      # (a - b) in units of the tolerances
        return torch.cat([(a["image"].double() - b["image"].double()) / tol[0],
                          (a["semantics"].double() - b["semantics"].double()) / tol[1],
                          ((a["depth"].double() - b["depth"].double()) /
                           (tol[2] * b["depth"].double().abs().clamp_min(1e-3))).view(1)])

    base = outputs(None)
    g = vec(got, base)
    for j in torch.argsort(impact, descending=True)[:top].tolist():
        h = JITTER * float(dz[j])
        eps, best_j = 0.0, None
        for _ in range(2):       # linear estimate around eps, then once more around the result
            here = base if eps == 0.0 else outputs({j: eps})
            probe = eps + (0.5 * h if eps <= 0 else -0.5 * h)
            J = vec(outputs({j: probe}), here) / (probe - eps)
            step = float((J @ vec(got, here)) / (J @ J).clamp_min(1e-300))
            eps = min(max(eps + step, -h), h)
            alt = outputs({j: eps})
            ei, es, ed = _errors(got, alt)
            cand = (max(ei / tol[0], es / tol[1], ed / tol[2]), j, eps, eps / max(float(dz[j]), 1e-30),
                    (ei, es, ed))
            if best_j is None or cand[0] < best_j[0]:
                best_j = cand
        yield best_j


def flagged_a_priori(aux, floor=WINDOW_MIN):
    """Per ray, BEFORE looking at any result: does it have a candidate at all
    (a weight inside the mask window, or a fine sample on the denom step)?"""
    w = aux["weights"]
    at_mask = ((w - 1e-4).abs() <= mask_window(aux, floor)).any(-1)
    if "w_coarse" in aux:
        denom, _ = fine_sample_cdf(aux)
        at_denom = ((denom - 1e-5).abs() <= DENOM_WINDOW).any(-1)
    else:
        at_denom = torch.zeros_like(at_mask)
    return at_mask, at_denom


MAX_BY_JITTER = 2   # rays of ONE check_render call explained by the moved-sample family


def check_render(res, ref, fld, rays, aabb, T, t, sel=None, tag="", t_rand=None,
                 max_loose_frac=5e-3, collect_unexplained=None, jitter=False,
                 max_by_jitter=MAX_BY_JITTER):
    """``res``: the outputs under test ([1, N(, C)] tensors, any device);
    ``ref``: ``oracle.renderer.run(..., return_aux=True)`` on ``rays`` = (o, d,
    nrm) [1, n, .] CPU tensors (the rows ``sel`` of what ``res`` rendered).

    ``collect_unexplained`` (a list, whole-view test only): loose rays that match
    no alternative are appended to it as (line, residuals, errors) instead of
    failing here -- the caller bounds their number and size over the whole view
    (one view has ~550 loose rays: a handful of them lie in the tail of the
    ordinary error distribution the match tolerance is a percentile of).

    ``max_by_jitter``: with ``jitter=True`` at most this many rays of the call may be
    explained by the moved-fine-sample family (ADVICE r5 / VERDICT r5: a one-parameter
    family, so its use is COUNTED and BOUNDED in every caller -- 2 per render sample
    here; a whole 307 200-ray view has shown 0-3 and bounds them itself, passing
    ``None``)."""
    o, d, nrm = rays[0].reshape(-1, 3), rays[1].reshape(-1, 3), rays[2].reshape(-1)
    aux = ref["aux"]
    pick = (lambda x: x[0].cpu()) if sel is None else (lambda x: x[0][sel.to(x.device)].cpu())
    got = {k: pick(res[k]) for k in ("image", "semantics", "depth")}
    e_img = (got["image"].double() - ref["image"][0].double()).abs().max(-1)[0]
    e_sem = (got["semantics"].double() - ref["semantics"][0].double()).abs().max(-1)[0]
    rel = ((got["depth"].double() - ref["depth"][0].double()).abs() /
           ref["depth"][0].double().abs().clamp_min(1e-3))
    n = rel.numel()
    for name, e, tol, cap in (("image", e_img, TOL_ABS, CAP_ABS), ("semantics", e_sem, TOL_ABS, CAP_ABS),
                              ("depth(rel)", rel, TOL_DEPTH_REL, CAP_DEPTH_REL)):
        print(f"{tag} {name}: median {float(e.median()):.2e} p99.5 {float(e.quantile(0.995)):.2e} "
              f"max {float(e.max()):.2e}; {int((e > tol).sum())} of {n} rays above {tol:g}")
        assert float(e.max()) <= cap, (tag, name, float(e.max()))
        assert float(e.median()) <= 5e-6, (tag, name, float(e.median()))
    loose = (e_img > TOL_ABS) | (e_sem > TOL_ABS) | (rel > TOL_DEPTH_REL)
    floor = window_floor(e_img, loose)
    # the floor follows the render's own median error, so it is capped by
    # a CONSTANT (2 % of the threshold, the largest window the rule ever had) and the
    # median itself is held to 5e-6 above: a regression cannot widen its own excuse
    # beyond what a healthy render already gets
    assert WINDOW_MIN <= floor <= WINDOW_MAX, floor
    at_mask, at_denom = flagged_a_priori(aux, floor)
    frac = float((at_mask | at_denom).float().mean())
    print(f"{tag} a-priori candidates (window floor {floor:.1e}): {int(at_mask.sum())} rays with a weight in the mask window, "
          f"{int(at_denom.sum())} with a fine sample on the denom step = {100 * frac:.1f} % of {n} "
          "(informational: having a candidate excuses nothing)")
    # every loose ray must match one alternative of ITS OWN candidates
    alt = alt_tolerances(e_img, e_sem, rel, loose)
    print(f"{tag} match tolerances (image, semantics, depth rel): {alt[0]:.2e} {alt[1]:.2e} {alt[2]:.2e}")
    # the match tolerance follows the render's own ordinary error, so
    # a regression of that error must not be able to loosen it without bound --
    # it never exceeds HALF the stated tolerance (a constant), and the ordinary
    # error itself is held by the median / loose-count assertions here
    assert alt[0] <= 0.5 * TOL_ABS and alt[1] <= 0.5 * TOL_ABS and alt[2] <= 0.5 * TOL_DEPTH_REL, alt
    unexplained = []
    n_jitter = 0
    for i in torch.nonzero(loose).flatten().tolist():
        ro = RayOracle(fld, o[i], d[i], nrm[i], aabb, T, t,
                       None if t == 0 else aux["u"][i],
                       None if t_rand is None else t_rand[i])
        g1 = {k: got[k][i] for k in got}
        score, what, errs, n_alt = explain_ray(ro, g1, tol=alt, floor=floor, jitter=jitter)
        line = (f"{tag} ray {i}: err img {float(e_img[i]):.2e} sem {float(e_sem[i]):.2e} depth "
                f"{float(rel[i]):.2e}; best of {n_alt} alternatives {what} -> "
                f"img {errs[0]:.2e} sem {errs[1]:.2e} depth {errs[2]:.2e}")
        print(line)
        if score <= 1.0 and what.get("moved_fine_sample"):
            n_jitter += 1
        if not (score <= 1.0 and (what["denom_flipped"] or what["mask_toggled"] or what.get("moved_fine_sample"))):
            if collect_unexplained is not None:
                collect_unexplained.append((line, errs, (float(e_img[i]), float(e_sem[i]), float(rel[i]))))
            else:
                unexplained.append(line)
    assert not unexplained, ("rays above the tolerance that NO alternative decision of the "
                             "reference's two step functions reproduces within "
                             f"{alt[0]:.2g} / {alt[1]:.2g} / {alt[2]:.2g}:\n" + "\n".join(unexplained[:10]))
    # every loose ray is reproduced by a named alternative above; their NUMBER
    # stays small: at most 0.5 % of the rays -- of the POPULATION the render samples.
    # A whole view of a field that leaves haze in empty space has 0.42 % (1295 of
    # 307 200); 4096 of its rays then hold 17 +- 4, and a fixed 20 would fail one
    # such sample in five.  So: the count must be consistent with 0.5 % at three
    # standard deviations of the binomial (4096 rays: 34; 32 768: 202; never below 8:
    # a 512-ray sample showed 5 once).
    # From 32 768 rays on the sample IS a population: 0.5 % is a hard limit there
    # (ADVICE r5).
    p_n = max_loose_frac * n
    limit = int(p_n) if n >= 32768 else max(8, int(math.ceil(p_n + 3.0 * math.sqrt(p_n))))
    assert int(loose.sum()) <= limit, (int(loose.sum()), limit)
    if max_by_jitter is not None:
        assert n_jitter <= max_by_jitter, (
            f"{tag}: {n_jitter} rays needed the moved-fine-sample alternative (bound {max_by_jitter})")
    return {"loose": int(loose.sum()), "flagged_frac": frac, "by_jitter": n_jitter}

"""The BASELINE configurations AT THEIR WORKLOADS, HIP path vs the CPU oracle
(``-m gpu``):

* cfg1 -- 4096 rays (a 64x64 crop), 16+16 samples: whole batch vs oracle.
* cfg2 -- the EXACT path ``bench.py`` times: one 640x480 view,
  ``render(staged=True, image_width=640, rng_u=...)`` on the bench's own field
  (``bench.build_field``: seed 123 + 200 Adam steps, trained in deterministic
  mode so that every box tests the same field: tests/util.bench_field), i.e.
  ``k_hashgrid_encode_tiled`` on 96-row bands of 61 440 rays; 4096 randomly
  picked pixels are compared with ``oracle.renderer.run`` on those rays and
  their rows of ``u`` (rays are independent).  Also with the lively
  U(-3,3) field, where nearly every sample passes the w > 1e-4 mask.
* cfg3 (NeRF half) -- reference-native 4096 rays x (256+256) samples,
  ``perturb=True``: loss values and the gradient of
  ``color + 0.04 sem + 0.1 depth`` (reference
  ``joint_train_lightning_net.py:188-223,503-507``) for all four parameter
  tensors against the oracle's autograd over the SAME 4096 rays (the oracle
  accumulates it over 4 ray blocks to bound host memory), plus a 256-ray
  case with a generic linear functional.

Tolerances as in test_gpu_parity / test_gpu_backward: image and semantics
1e-4 abs, depth 2e-4 rel (for >= 99.5 % of the rays at these sizes; every
ray above must be reproduced by a specific alternative decision of the
reference's own step functions, see ``_check`` / tests/parity_check.py), gradients 2e-3 relative L2 (5e-3 for the hash grid at cfg3 size:
~2 M samples scatter into it through float atomics)."""
import numpy as np
import pytest
import torch

from oracle import field as ofield
from oracle import losses as olosses
from oracle import rays as orays
from oracle import renderer as oren
from tests import parity_check as pc
from tests.util import (AABB4, bench_field, hip_network_from_oracle, lively_oracle_field,
                        make_rays, maxabs)

pytestmark = pytest.mark.gpu


def _oracle_from_net(net):
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=net.num_semantic_classes,
                             seed=None)
    fld.grid_params = net.encoder.params.detach().cpu().clone()
    fld.sigma_params = net.sigma_net.params.detach().cpu().clone()
    fld.color_params = net.color_net.params.detach().cpu().clone()
    fld.sem_params = net.semantics_net.params.detach().cpu().clone()
    return fld


def _check(res, ref, fld, rays, T, t, sel=None, tag=""):
    """Stated tolerance: image / semantics 1e-4 abs, depth 2e-4 rel -- for at
    least 99.5 % of the rays; every ray within 2e-3 / 5e-3; median <= 5e-6.
    EVERY ray above 1e-4 / 2e-4 must match, in all three outputs at once and
    within twice the render's own p99.5 error over its ordinary rays (at least
    2e-5 / 5e-5 rel, never more than 5e-5 / 1e-4 rel = half the stated tolerance:
    parity_check.alt_tolerances), the ORACLE re-evaluated on that ray with one of
    its at-threshold decisions taken the other way (mask bits of the <= 3
    samples whose weight is within fp32 noise of 1e-4; the branch of the <= 2
    fine samples on sample_pdf's ``denom < 1e-5`` step) -- tests/parity_check.py,
    itself tested by tests/test_parity_check_cpu.py.  No blanket escape.
    Since the end of round 5 also: ONE fine sample moved by <= 6 x the modelled
    round-off of its depth with the field re-evaluated there (the oracle against
    itself with its cdf summed in another order needs this on 2 of 4096 rays:
    test_the_oracle_with_its_cdf_summed_in_another_order_is_fully_explained; at
    16+16 samples, where a bin is 0.4 wide, most loose rays are of this kind:
    tests/scripts/checker_flake_rate.py with TT=16).  Their number is reported
    (`by_jitter`) and bounded: at most parity_check.MAX_BY_JITTER = 2 per call here
    (asserted in check_render), at most 8 of a whole view's 307 200."""
    return pc.check_render(res, ref, fld, rays, AABB4, T, t, sel=sel, tag=tag, jitter=True)


def test_cfg1_4096_rays_16_plus_16():
    fld = lively_oracle_field()
    net = hip_network_from_oracle(fld).eval()
    # a 64x64 crop of a 640x480 camera
    from oracle.rays import pixel_rays_train
    pose = torch.eye(4)[None].clone()
    pose[0, :3, 3] = torch.tensor([0.3, -0.2, 0.1])
    yy, xx = torch.meshgrid(torch.arange(200, 264), torch.arange(300, 364), indexing="ij")
    inds = (yy * 640 + xx).reshape(-1)
    o, d, nrm, _ = pixel_rays_train(pose, (569.6, 569.6, 320.0, 240.0), 480, 640, inds)
    g = torch.Generator().manual_seed(1)
    u = torch.rand(4096, 16, generator=g)
    with torch.no_grad():
        ref = oren.run(fld, o, d, nrm, AABB4, num_steps=16, upsample_steps=16, u=u,
                       return_aux=True)
        res = net.render(o.cuda(), d.cuda(), nrm.cuda(), staged=True, num_steps=16,
                         upsample_steps=16, rng_u=u.cuda())
    _check(res, ref, fld, (o, d, nrm), 16, 16, tag="cfg1")


def test_cfg2_whole_view_f16x2_every_ray_against_the_oracle():
    """the test below compares 4 096 of a view's 307 200
    rays.  Here EVERY ray of one 640x480 view rendered exactly as bench.py times
    it (f16x2 nets, one pipelined ucsa_render_view call, depth-ordered fine pass)
    goes against the oracle, in blocks of 32 768 rays (~6 s of CPU each), with the
    same stated tolerance and the same causal explanation of the rays above it
    (a decision of one of the two step functions, or one fine
    sample moved inside its depth round-off; all but a bounded handful of a view's
    ~550: see the end of the test)."""
    import bench
    from ucsa_neural_rendering_amd import ops
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
    dev = torch.device("cuda:0")
    H, W, T, t = bench.H, bench.W, bench.T_COARSE, bench.T_FINE
    net, _ = bench_field(dev)      # deterministic training: the same field on every box
    fld = _oracle_from_net(net)
    net.hip_ray_chunk = 65536
    net.precision = "f16x2"
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    pose = _slerp_loop_poses(23, seed=999)[11:12].to(dev)
    o, d, nrm = ops.get_rays(pose, intr, H, W)
    g = torch.Generator(device=dev).manual_seed(1001)
    u = torch.rand(H * W, t, device=dev, generator=g)
    with torch.no_grad():
        res = net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                         upsample_steps=t, rng_u=u, image_width=W)
    torch.cuda.synchronize()
    oc, dc, nc, uc = o.cpu(), d.cpu(), nrm.cpu(), u.cpu()
    loose = n = moved = 0
    tail = []
    for head in range(0, H * W, 32768):
        sel = torch.arange(head, min(head + 32768, H * W))
        rays_sel = (oc[:, sel], dc[:, sel], nc[:, sel])
        with torch.no_grad():
            ref = oren.run(fld, *rays_sel, AABB4, num_steps=T, upsample_steps=t, u=uc[sel],
                           return_aux=True)
        # (the 99.5 % criterion is applied to the VIEW below, not to each block: on
        # one of 22 fields a block had 170 loose rays = 0.52 %, nearly all of them single
        # mask flips of a sample with class probability 1.0 -- an error of exactly the
        # threshold weight 1.0e-4, i.e. the stated tolerance itself)
        r = pc.check_render(res, ref, fld, rays_sel, AABB4, T, t, sel=sel,
                            tag=f"cfg2-whole[{head}]", collect_unexplained=tail,
                            max_loose_frac=1.0, jitter=True, max_by_jitter=None)
        loose += r["loose"]
        moved += r["by_jitter"]
        n += sel.numel()
    print(f"cfg2 whole view: {loose} of {n} rays above the stated tolerance; {loose - len(tail)} of them "
          f"reproduced by a named alternative within the match tolerance ({moved} by one fine sample moved "
          f"inside its depth round-off, the others by a decision of a step function), {len(tail)} not:")
    for line, resid, errs in tail:
        print("   TAIL " + line)
    assert n == H * W
    assert loose <= 0.005 * n, (loose, n)     # >= 99.5 % of the view's rays within the stated tolerance
    # The handful that matches no alternative (profiles/r05_whole_view_tail.txt, 40
    # fields): 0-2 rays per view, of two kinds --
    # (i) marginal: a mask flip accounts for the ray down to a residual of 2.2e-5 ...
    # 3.5e-5 against a match tolerance of 2.0e-5 ... 3.2e-5 (the tolerance is a
    # PERCENTILE of the view's ordinary error: of ~550 loose rays one in its tail is
    # expected every second view), or the flipped sample sat 1.3 % from the threshold
    # against a window of 1.29 %;
    # (ii) semantics-only residuals of 7e-5 ... 1.7e-4 next to 3e-6 in image and
    # depth: ONE fine sample next to a class boundary sits a few times its depth
    # round-off away (a sample drawn into a nearly empty bin is placed by the ratio
    # of two cdf differences: 4 ulp of the running sum are ~2 % of the bin); the
    # weights hardly move, the class probabilities AT the sample do.  Reproduced
    # without a GPU (tests/scripts/oracle_self_noise.py: the oracle against itself
    # with another summation order and 1e-5 of density noise shows two such rays on
    # a view) and matched by parity_check._moved_fine_sample -- that sample moved by
    # <= 6 dz, the field re-evaluated there -- to 9e-7; counted and bounded below.
    # Until the window had a floor at the render's own error level (parity_check
    # .window_floor, round 5) a third kind existed: on fields with ~1200 loose rays
    # (haze: many weights of 1e-4 behind empty space) up to 7 single flips of a
    # p = 1.0 sample, error 1.00e-4 ... 1.02e-4, had NO candidate in the depth-only
    # window -- that is what failed 2 runs in 5 of the suite.  With the floor: 0 or 1
    # tail ray on 8 further views, two of them such fields (1239 / 842 loose rays).
    # Bound: at most 8 rays = 2.6e-5 of the view (or 1 % of the loose rays), each inside the hard cap that
    # check_render asserts for EVERY ray, every one of them printed.
    assert len(tail) <= max(8, loose // 100), [x[0] for x in tail]      # (a haze field: ~1300 loose rays, 4 seen)
    assert moved <= 8, moved      # the odd ray, not a second tolerance
    for line, resid, errs in tail:
        assert errs[0] <= pc.CAP_ABS and errs[1] <= pc.CAP_ABS and errs[2] <= pc.CAP_DEPTH_REL, line


@pytest.mark.parametrize("which", ["bench_field", "lively_field"])
def test_cfg2_bench_path_640x480_staged_image_ordered(which):
    import bench
    from ucsa_neural_rendering_amd import ops
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
    dev = torch.device("cuda:0")
    H, W, T, t = bench.H, bench.W, bench.T_COARSE, bench.T_FINE
    if which == "bench_field":
        net, _ = bench_field(dev)      # deterministic training: the same field on every box
        fld = _oracle_from_net(net)
    else:
        fld = lively_oracle_field()
        net = hip_network_from_oracle(fld).eval()
    net.hip_ray_chunk = 65536                       # as bench.main sets it
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    pose = _slerp_loop_poses(23, seed=999)[7:8].to(dev)
    o, d, nrm = ops.get_rays(pose, intr, H, W)
    g = torch.Generator(device=dev).manual_seed(1000)
    u = torch.rand(H * W, t, device=dev, generator=g)
    with torch.no_grad():
        res = net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                         upsample_steps=t, rng_u=u, image_width=W)
    torch.cuda.synchronize()
    for k in ("image", "depth", "semantics"):
        assert torch.isfinite(res[k]).all(), k
    # 4096 random pixels, across every 96-row band and the ragged last one
    gs = torch.Generator().manual_seed(4)
    sel = torch.randperm(H * W, generator=gs)[:4096]
    sel[:8] = torch.tensor([0, W - 1, 96 * W - 1, 96 * W, 5 * 96 * W - 1,
                            H * W - W, H * W - 1, 384 * W + 17])
    rays_sel = (o.cpu()[:, sel], d.cpu()[:, sel], nrm.cpu()[:, sel])
    with torch.no_grad():
        ref = oren.run(fld, *rays_sel, AABB4,
                       num_steps=T, upsample_steps=t, u=u.cpu()[sel], return_aux=True)
    _check(res, ref, fld, rays_sel, T, t, sel, tag=f"cfg2[{which}]")
    # round 3's default arithmetic (bf16x3, fp32-grade on the bf16 MFMA pipe):
    # the same fp32 tolerances against the oracle, and 1e-6 to the exact chain
    net.precision = "bf16x3"
    with torch.no_grad():
        res3 = net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                          upsample_steps=t, rng_u=u, image_width=W)
    net.precision = "fp32"
    _check(res3, ref, fld, rays_sel, T, t, sel, tag=f"cfg2[{which}, bf16x3]")
    # ... and bench.py's default since round 4: f16x2 (two-term f16 operands, half
    # the matrix passes, the same fp32-grade error)
    net.precision = "f16x2"
    with torch.no_grad():
        resh = net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                          upsample_steps=t, rng_u=u, image_width=W)
    net.precision = "fp32"
    _check(resh, ref, fld, rays_sel, T, t, sel, tag=f"cfg2[{which}, f16x2]")
    for k in ("image", "semantics"):
        e = (resh[k][0] - res[k][0]).abs().max(-1)[0]
        print(f"cfg2[{which}] f16x2 vs f32 MFMA, {k}: median {float(e.median()):.2e} "
              f"p99 {float(e.quantile(0.99)):.2e} p99.9 {float(e.quantile(0.999)):.2e} "
              f"max {float(e.max()):.2e}")
        assert float(e.median()) <= 2e-6 and float(e.quantile(0.999)) <= 2e-4 \
            and float(e.max()) <= 2e-3, k
    for k in ("image", "semantics"):
        # The nets of the two modes agree to ~1e-7 (test_bf16x3_nets_are_fp32_
        # grade); a whole view also passes the two step functions of the path
        # (mask w > 1e-4, sample_pdf's denom < 1e-5, see _check): a 1e-6
        # relative difference in sigma moves a fine sample across an empty bin
        # on a fraction of a percent of the rays.  Hence statistics.
        e = (res3[k][0] - res[k][0]).abs().max(-1)[0]
        print(f"cfg2[{which}] bf16x3 vs f32 MFMA, {k}: median {float(e.median()):.2e} "
              f"p99 {float(e.quantile(0.99)):.2e} p99.9 {float(e.quantile(0.999)):.2e} "
              f"max {float(e.max()):.2e}")
        # 307 200 rays: the bulk agrees to round-off, a ray on a mask step
        # moves by <= 1e-4 per flipped sample, one on the denom step by up to
        # _check's hard cap
        assert float(e.median()) <= 2e-6 and float(e.quantile(0.999)) <= 2e-4 \
            and float(e.max()) <= 2e-3, k
    # the fp16-MFMA option on the same path, against the oracle emulating
    # tcnn's roundings (fp16 weights / layer inputs, fp32 accumulate)
    import copy
    f16 = copy.copy(fld)
    f16.emulate_fp16 = True
    net.precision = "fp16"
    with torch.no_grad():
        res16 = net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                           upsample_steps=t, rng_u=u, image_width=W)
        sub = sel[:1024]
        ref16 = oren.run(f16, o.cpu()[:, sub], d.cpu()[:, sub], nrm.cpu()[:, sub], AABB4,
                         num_steps=T, upsample_steps=t, u=u.cpu()[sub])
    net.precision = "fp32"
    for k in ("image", "semantics"):
        # 3e-3 for fp16 arithmetic; the two step functions on top of it on the
        # rare ray that sits on one (hard cap 6e-3)
        e16 = (res16[k][0][sub.to(dev)].cpu() - ref16[k][0]).abs().max(-1)[0]
        print(f"cfg2[{which}] fp16 vs emulating oracle, {k}: p99.5 "
              f"{float(e16.quantile(0.995)):.2e} max {float(e16.max()):.2e}")
        assert float(e16.quantile(0.995)) <= 3e-3 and float(e16.max()) <= 6e-3, k


def test_cfg4_512_views_on_one_gpu_with_oracle_spot_checks():
    """BASELINE cfg4 AT ITS SIZE as far as one GPU allows: all 512 novel
    640x480 views (157 M rays, 192 samples each) through ``bench.cfg4_job`` --
    the function ``bench.py --mode cfg4`` times, world 1 -- on the bench's own
    field; five of the views, spread over the job (first, last and three in
    between), are kept and 512 random pixels of each are compared with
    ``oracle.renderer.run`` on the same rays and rows of ``u``.  On 8 GPUs
    each rank runs the same loop over views rank, rank+8, ... (no data-path
    collective), so what is untested here is only that eight processes run
    side by side (tests/test_gpu_bench_modes.py covers two)."""
    import bench
    dev = torch.device("cuda:0")
    net, _ = bench_field(dev)      # deterministic training: the same field on every box
    net.hip_ray_chunk = 65536
    fld = _oracle_from_net(net)
    keep = (0, 101, 257, 389, 511)
    elapsed, mine, kept = bench.cfg4_job(net, 512, 0, 1, dev, keep=keep,
                                         precision="f16x2")   # what --mode cfg4 runs
    assert mine == list(range(512)) and sorted(kept) == list(keep)
    print(f"cfg4: 512 views in {elapsed:.2f} s = {512 * 480 * 640 / elapsed / 1e6:.2f} M rays/s")
    assert elapsed < 60.0            # ~12 s expected; a stall would show here
    for v in keep:
        rec = kept[v]
        for k in ("image", "depth", "semantics"):
            assert torch.isfinite(rec[k]).all(), (v, k)
        gs = torch.Generator().manual_seed(100 + v)
        sel = torch.randperm(480 * 640, generator=gs)[:512]
        rays_sel = (rec["o"].cpu()[:, sel], rec["d"].cpu()[:, sel], rec["nrm"].cpu()[:, sel])
        with torch.no_grad():
            ref = oren.run(fld, *rays_sel, AABB4, num_steps=bench.T_COARSE,
                           upsample_steps=bench.T_FINE, u=rec["u"].cpu()[sel],
                           return_aux=True)
        _check(rec, ref, fld, rays_sel, bench.T_COARSE, bench.T_FINE, sel,
               tag=f"cfg4[view {v}]")
    # distinct poses gave distinct images
    assert maxabs(kept[0]["image"], kept[257]["image"]) > 1e-2


def _rel_l2(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


@pytest.mark.parametrize("train_precision", ["fp32", "bf16x3"])
def test_cfg3_gradients_256_rays_256_plus_256_perturbed(train_precision):
    """``bf16x3``: the training forward's colour / semantics stage on the split
    pair (ucsa_composite_train_fwd_x3), same tolerances."""
    N, T, t = 256, 256, 256
    fld = lively_oracle_field().requires_grad_(True)
    net = hip_network_from_oracle(fld).train()
    net.train_precision = train_precision
    o, d, norms = make_rays(N, 901)
    g = torch.Generator().manual_seed(N)
    t_rand, u = torch.rand(N, T, generator=g), torch.rand(N, t, generator=g)
    ci, cd, cs = (torch.rand(1, N, 3, generator=g), torch.rand(1, N, generator=g),
                  torch.rand(1, N, 40, generator=g))
    ref = oren.run(fld, o[None], d[None], norms[None], AABB4, num_steps=T,
                   upsample_steps=t, t_rand=t_rand, u=u)
    ((ref["image"] * ci).sum() + (ref["depth"] * cd).sum() + (ref["semantics"] * cs).sum()).backward()
    res = net.render(o[None].cuda(), d[None].cuda(), norms[None].cuda(), perturb=True,
                     num_steps=T, upsample_steps=t, rng_t=t_rand.cuda(), rng_u=u.cuda())
    assert maxabs(res["image"], ref["image"]) <= 1e-4
    assert maxabs(res["semantics"], ref["semantics"]) <= 1e-4
    ((res["image"] * ci.cuda()).sum() + (res["depth"] * cd.cuda()).sum()
     + (res["semantics"] * cs.cuda()).sum()).backward()
    for name, got, want in (("color", net.color_net.params.grad, fld.color_params.grad),
                            ("sem", net.semantics_net.params.grad, fld.sem_params.grad),
                            ("sigma", net.sigma_net.params.grad, fld.sigma_params.grad),
                            ("grid", net.encoder.params.grad, fld.grid_params.grad)):
        print(f"cfg3-256 {name}: rel L2 {_rel_l2(got, want):.3e}")
        assert _rel_l2(got, want) <= 2e-3, name


@pytest.mark.parametrize("train_precision", ["fp32", "bf16x3"])
def test_cfg3_train_step_4096_rays_512_samples_loss_and_gradients(train_precision):
    """The reference-native NeRF training batch in one HIP call vs the oracle
    over the same rays.  The loss terms are means over the batch (depth: over
    the valid pixels), so the oracle's value / gradient over 4 blocks of 1024
    rays are the block terms weighted by n_block/N (valid_block/valid)."""
    from ucsa_neural_rendering_amd import losses as ul
    N, T, t, C = 4096, 256, 256, 40
    fld = lively_oracle_field().requires_grad_(True)
    net = hip_network_from_oracle(fld).train()
    net.train_precision = train_precision
    o, d, norms = make_rays(N, 77)
    g = torch.Generator().manual_seed(7)
    t_rand, u = torch.rand(N, T, generator=g), torch.rand(N, t, generator=g)
    gt_rgb = torch.rand(1, N, 3, generator=g)
    labels = torch.randint(-1, C, (1, N), generator=g)
    gt_depth = torch.rand(1, N, generator=g) * 4 + 0.3
    gt_depth[0, ::11] = 0                         # invalid depth pixels
    uom = 0.9

    res = net.render(o[None].cuda(), d[None].cuda(), norms[None].cuda(), perturb=True,
                     num_steps=T, upsample_steps=t, rng_t=t_rand.cuda(), rng_u=u.cuda())
    lc, ls, ld = ul.nerf_losses(res["image"], res["semantics"], res["depth"],
                                gt_rgb.cuda(), labels.cuda(), gt_depth.cuda(), uom)
    ul.nerf_total_loss(lc, ls, ld).backward()
    torch.cuda.synchronize()

    n_valid = int((gt_depth != 0).sum())
    tot = dict(c=0.0, s=0.0, d=0.0)
    img_err = 0.0
    for head in range(0, N, 1024):
        sl = slice(head, head + 1024)
        ref = oren.run(fld, o[None, sl], d[None, sl], norms[None, sl], AABB4, num_steps=T,
                       upsample_steps=t, t_rand=t_rand[sl], u=u[sl])
        img_err = max(img_err, maxabs(res["image"][:, sl], ref["image"]))
        rc, rs, rd = olosses.nerf_losses(ref["image"], ref["semantics"], ref["depth"],
                                         gt_rgb[:, sl], labels[:, sl], gt_depth[:, sl], uom)
        wv = int((gt_depth[:, sl] != 0).sum()) / n_valid
        (rc * (1024 / N) + rs * (1024 / N) * olosses.WEIGHT_SEMANTICS
         + rd * wv * olosses.WEIGHT_DEPTH).backward()
        tot["c"] += float(rc) * 1024 / N
        tot["s"] += float(rs) * 1024 / N
        tot["d"] += float(rd) * wv
    assert img_err <= 1e-4
    print(f"cfg3-4096 losses hip {float(lc):.6f} {float(ls):.6f} {float(ld):.6f} "
          f"oracle {tot['c']:.6f} {tot['s']:.6f} {tot['d']:.6f}")
    assert abs(float(lc) - tot["c"]) <= 1e-5 * max(1.0, abs(tot["c"]))
    assert abs(float(ls) - tot["s"]) <= 1e-5 * max(1.0, abs(tot["s"]))
    assert abs(float(ld) - tot["d"]) <= 1e-5 * max(1.0, abs(tot["d"]))
    for name, got, want, tol in (
            ("color", net.color_net.params.grad, fld.color_params.grad, 2e-3),
            ("sem", net.semantics_net.params.grad, fld.sem_params.grad, 2e-3),
            ("sigma", net.sigma_net.params.grad, fld.sigma_params.grad, 2e-3),
            ("grid", net.encoder.params.grad, fld.grid_params.grad, 5e-3)):
        print(f"cfg3-4096 {name}: rel L2 {_rel_l2(got, want):.3e} "
              f"|g| {float(want.norm()):.3e}")
        assert _rel_l2(got, want) <= tol, name
